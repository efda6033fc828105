#!/usr/bin/env python3
"""BASELINE.json configs[4] (and the reference's own golden table, basisu_tool.cpp:6737-6776) on their REAL inputs: kodim01..24.

Writes
  tests/golden/kodak24.npz            the 24 images (RGB u8, 768x512 or 512x768) -- a fixture of the reference (test_files/kodimNN.png)
                                      that the GPU box does not have
  tests/golden/kodak24_digests.json   per image, from the real reference (oracle/_ref, single-threaded = the pinned configuration):
      uastc_l2                 sha256 of encode_uastc level 2 over all blocks (basisu -uastc -uastc_level 2)
      uastc_l2_rdo1_jobs{1,4}  sha256 after uastc_rdo, lambda 1.0, default parameters, 1 strip (-no_multithreading) / 4 strips (the tool's
                               min(4, threads), comp.cpp:2078)
      uastc_l0 (+ psnr)        the same at level 0, which is what the reference's -test mode encodes (basis_compress with no level bits set)
      uastc_psnr_rgba[_rdo]    image_metrics RGBA PSNR of the reference's decoded blocks (unpack_uastc) -- what basisu -stats prints
      etc1s_q128               frontend state digests, backend payload digests, RGBA PSNR of the decoded output and size + sha256 of the file
                               `basisu -etc1s -q 128 -comp_level 1 -no_multithreading` writes, with the key-values it stores
      etc1s_q128_table         the same for `-comp_level 2 -linear` = the library defaults the reference's -test mode encodes with
  plus "reference_table": the rows of g_etc1s_uastc_4x4_ldr_test_files (sizes / PSNRs its own -test mode accepts within 4.5 % / 0.3 dB).
Run in the build container (needs /root/reference/test_files and oracle/_ref); ~5 minutes."""
import hashlib
import json
import pathlib
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import helpers  # noqa: E402
import test_gpu_etc1s_frontend as T  # noqa: E402
from basis_universal_amd.etc1s import quality_to_clusters  # noqa: E402
from basis_universal_amd.backend import default_params  # noqa: E402

OUT = ROOT / "tests" / "golden" / "kodak24_digests.json"
NPZ = ROOT / "tests" / "golden" / "kodak24.npz"
PAYLOAD = ("endpoint_palette", "selector_palette", "slice_image_tables", "slice_image_data", "slice_image_crcs")

# basisu_tool.cpp:6746-6769: (etc1s q1 size, etc1s q1 psnr, uastc psnr, etc1s q128 size, etc1s q128 psnr), RGBA-average PSNRs
REFERENCE_TABLE = {
    1: (31003, 27.40, 44.14, 58385, 30.356064), 2: (28560, 32.20, 41.06, 51442, 34.713940), 3: (23442, 32.57, 44.87, 49548, 36.709675),
    4: (28287, 31.76, 43.02, 57034, 34.864861), 5: (32677, 25.94, 40.28, 65742, 29.935091), 6: (27367, 28.66, 44.57, 54994, 32.294220),
    7: (26649, 31.51, 43.94, 53374, 35.576595), 8: (31164, 25.28, 41.15, 63516, 29.509914), 9: (24808, 32.05, 45.85, 51402, 35.985966),
    10: (27278, 32.20, 45.77, 54322, 36.395000), 11: (26610, 29.22, 43.68, 55526, 33.468971), 12: (25133, 32.96, 46.77, 51503, 36.722233),
    13: (31635, 24.25, 41.25, 62660, 27.588623), 14: (31193, 27.81, 39.65, 62897, 31.206463), 15: (25559, 31.26, 42.87, 53424, 35.026314),
    16: (26925, 32.21, 47.78, 51354, 35.555458), 17: (29365, 31.40, 45.66, 55675, 35.909283), 18: (30960, 27.46, 41.54, 62388, 31.348171),
    19: (27920, 29.69, 44.95, 55098, 33.613987), 20: (21135, 31.30, 45.31, 47160, 35.759407), 21: (25974, 28.53, 44.45, 54799, 32.415817),
    22: (29111, 29.85, 42.63, 60994, 33.495415), 23: (23825, 31.69, 45.11, 53614, 36.223492), 24: (29644, 26.75, 40.61, 58909, 31.522869),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_decode_uastc(packed, nbx, nby):
    import ctypes as C
    L = helpers.ref()
    L.ref_unpack_uastc.restype = C.c_int
    L.ref_unpack_uastc.argtypes = [helpers.u8p, helpers.u8p]
    out = np.zeros((packed.shape[0], 4, 4, 4), np.uint8)
    for i in range(packed.shape[0]):
        assert L.ref_unpack_uastc(helpers.ptr(packed[i]), helpers.ptr(out[i])) == 1
    return out.reshape(nby, nbx, 4, 4, 4).transpose(0, 2, 1, 3, 4).reshape(nby * 4, nbx * 4, 4)


def main():
    images = {}
    for k in range(1, 25):
        rgba = helpers.load_png(helpers.REF_DIR / "test_files" / f"kodim{k:02d}.png")
        assert (rgba[..., 3] == 255).all()
        images[f"k{k:02d}"] = np.ascontiguousarray(rgba[..., :3])
    np.savez_compressed(NPZ, **images)
    out = {"reference_table": {f"k{k:02d}": dict(zip(("etc1s_q1_size", "etc1s_q1_psnr", "uastc_psnr", "etc1s_q128_size", "etc1s_q128_psnr"), v)) for k, v in REFERENCE_TABLE.items()},
           "images": {}}
    for name, rgb in images.items():
        t0 = time.time()
        h, w = rgb.shape[:2]
        rgba = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=2)
        blocks = helpers.to_pixel_blocks(rgba)
        nbx, nby = w // 4, h // 4
        rec = {"width": w, "height": h, "n_blocks": int(blocks.shape[0]), "rgb_sha256": sha(rgb)}
        packed = helpers.ref_encode_uastc(blocks, 2)
        rec["uastc_l2"] = sha(packed)
        rec["uastc_psnr_rgba"] = round(helpers.psnr(ref_decode_uastc(packed, nbx, nby), rgba), 4)
        packed0 = helpers.ref_encode_uastc(blocks, 0)   # the level the reference's own -test mode runs (flags_and_quality & cPackUASTCLevelMask = 0)
        rec["uastc_l0"] = sha(packed0)
        rec["uastc_l0_psnr_rgba"] = round(helpers.psnr(ref_decode_uastc(packed0, nbx, nby), rgba), 4)
        for jobs in (1, 4):
            r = helpers.ref_uastc_rdo(packed, blocks, 2, 0 if jobs == 1 else jobs, lam=1.0)
            rec[f"uastc_l2_rdo1_jobs{jobs}"] = sha(r)
            rec[f"uastc_rdo1_jobs{jobs}_modified"] = int((r != packed).any(axis=1).sum())
            rec[f"uastc_psnr_rgba_rdo1_jobs{jobs}"] = round(helpers.psnr(ref_decode_uastc(r, nbx, nby), rgba), 4)
        # ETC1S -q 128 twice: the command line's defaults (comp level 1, sRGB metrics: BASELINE configs[0]'s settings) and what the reference's -test mode
        # runs through basis_compress(cETC1S, ..., 128 | threaded) = library defaults: comp level 2 (BASISU_DEFAULT_ETC1S_COMPRESSION_LEVEL), LINEAR metrics
        # (no cFlagSRGB: comp.cpp:5704-5708) -- the configuration its golden table's sizes / PSNRs belong to
        png = helpers.REF_DIR / "test_files" / f"kodim{name[1:]}.png"
        for key, level, perceptual, cli in (("etc1s_q128", 1, True, ("-comp_level", "1")), ("etc1s_q128_table", 2, False, ("-comp_level", "2", "-linear"))):
            max_ep, max_sel = quality_to_clusters(128, blocks.shape[0])
            fe = helpers.RefFrontend(blocks, max_ep, max_sel, level, perceptual)
            fe.call("compress")
            thr = tuple(float(x) for x in default_params(128, level))
            total, _ = fe.backend_run([(0, nbx, nby)], *thr)   # at levels > 1 the backend re-optimises the frontend's endpoints: state taken AFTER it
            st = {k: fe.get(k) for k in T.STATE}
            tool = helpers.run_ref_cli(png, "-etc1s", "-q", "128", *cli)
            prm = fe.get("endpoint_cluster_etc_params").reshape(-1, 16)[:, :4]
            eb = fe.backend_get("encoder_blocks")
            dec = helpers.decode_backend_blocks(eb, prm, fe.get("optimized_cluster_selectors"), nbx, nby)
            rec[key] = {
                "quality": 128, "level": level, "perceptual": perceptual, "max_endpoint_clusters": max_ep, "max_selector_clusters": max_sel,
                "frontend_digests_after_backend": T._digest(st),
                "backend": {"slices": [[0, nbx, nby]], "thresholds": list(thr), "compressed_bytes": int(total),
                            "digests": {k: sha(fe.backend_get(k)) for k in PAYLOAD}},
                "tool_basis_size": int(tool.size), "tool_basis_sha256": sha(tool),
                "tool_basis_key_values": [[k, bytes(v).hex()] for k, v in helpers.basis_file_key_values(tool)],
                "psnr_rgba": round(helpers.psnr(np.concatenate([dec, np.full((h, w, 1), 255, np.uint8)], axis=2), rgba), 4),
            }
            fe.close()
        out["images"][name] = rec
        print(name, w, h, rec["uastc_psnr_rgba"], rec["uastc_psnr_rgba_rdo1_jobs4"], rec["uastc_rdo1_jobs4_modified"], rec["etc1s_q128"]["tool_basis_size"], rec["etc1s_q128"]["psnr_rgba"], rec["etc1s_q128_table"]["tool_basis_size"], rec["etc1s_q128_table"]["psnr_rgba"], f"{time.time() - t0:.1f}s", flush=True)
        OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
