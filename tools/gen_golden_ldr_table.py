#!/usr/bin/env python3
"""The rows of the reference's own LDR golden tables that tools/gen_golden_kodak.py does not cover:

  * g_etc1s_uastc_4x4_ldr_test_files (basisu_tool.cpp:6737-6776, `basisu -test`) has 28 files -- kodim01..24 AND black_1x1, white_1x1, wikipedia
    (1845x894: neither dimension a multiple of 4, text edges), alpha0 (LA source -> colour + alpha slices) -- and THREE columns: ETC1S quality 1
    (m_etc1s_size / m_etc1s_psnr: basis_compress with no quality bits = max(1, 0), comp.cpp:5741), ETC1S quality 128, UASTC.
  * the ETC1S half of g_codec_test_cases (basisu_tool_test_codecs.inl:13-103, `basisu -test_codecs ETC1S`): kodim03 / 23 / 18, alpha0, wikipedia,
    black_1x1 x quality {10, 25, 50, 75, 100} x effort {0, 3, 6} -> .ktx2 size.
  * the UASTC LDR 4x4 half of the same grid (basisu_tool_test_codecs.inl:104-193, `basisu -test_codecs UASTC_LDR_4x4`): run_codec_test_case
    (basisu_tool.cpp:7704-7753) leaves m_ktx2_uastc_supercompression at its default KTX2_SS_NONE (comp.h:323), so these are plain UASTC level
    round(0.4 effort) + the RDO post-pass at lambda = 20 (1 - quality/100)^1.3 (quality 100: no post-pass), four strips, written without
    Zstandard: `basisu -ktx2 -ktx2_no_zstandard -uastc -quality Q -effort E` in the tool's default (multi-threaded) configuration.
    `python tools/gen_golden_ldr_table.py uastc_grid` makes only this section (~2 minutes).

Writes
  tests/golden/ldr_extra.npz          the four non-Kodak images as RGBA u8 (fixtures of the reference, test_files/*.png; the GPU box has none)
  tests/golden/ldr_table_digests.json per file and column: size + sha256 + key-values of the file the reference TOOL writes for the column's settings
                                      (-no_multithreading; all of these inputs are below the 262,144-vector gate of the multi-threaded codebook build, so
                                      the tool's default configuration writes the same bytes -- checked here for the largest one), the RGBA PSNR the tool
                                      prints for slice 0 (what image_stats::m_basis_rgba_avg_psnr holds), and for the four new files the block-level UASTC
                                      digests of gen_golden_kodak.py.
Run in the build container (needs /root/reference/test_files and oracle/_ref); ~6 minutes."""
import hashlib
import json
import pathlib
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import helpers  # noqa: E402

OUT = ROOT / "tests" / "golden" / "ldr_table_digests.json"
NPZ = ROOT / "tests" / "golden" / "ldr_extra.npz"
EXTRA = ("black_1x1", "white_1x1", "wikipedia", "alpha0")
# basisu_tool.cpp:6748, 6773-6775: (etc1s q1 size, etc1s q1 psnr, uastc psnr, etc1s q128 size, etc1s q128 psnr)
EXTRA_TABLE = {"black_1x1": (220, 100.0, 100.0, 220, 100.0), "white_1x1": (220, 100.0, 100.0, 220, 100.0),
               "wikipedia": (38992, 24.10, 30.47, 69608, 27.630802), "alpha0": (807, 100.0, 56.16, 810, 100.0)}
CODEC_FILES = ("kodim03", "kodim23", "kodim18", "alpha0", "wikipedia", "black_1x1")   # g_codec_ldr_test_files, basisu_tool.cpp:7654
QUALITIES, EFFORTS = (10, 25, 50, 75, 100), (0, 3, 6)                                 # :7656-7657


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def low_level(quality, effort):
    """basis_compressor_params::set_format_mode_and_quality_effort for cETC1S (comp.cpp:76-92, 158-176): std::round = half away from zero"""
    import math
    return int(math.floor(255.0 * quality / 100.0 + 0.5)), int(math.floor(6.0 * min(effort, 10) / 10.0 + 0.5))


def run_tool(png, *args, ktx2=False, multithreaded=False, all_psnrs=False):
    """-> (file bytes, the RGBA Avg PSNR the tool prints for slice 0); all_psnrs: (file bytes, {rgb, rgba, bc7_rgba})"""
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(png, pathlib.Path(d) / "in.png")
        cmd = [str(helpers.ORACLE_DIR / "_ref" / "basisu"), "-ktx2" if ktx2 else "-basis", *([] if multithreaded else ["-no_multithreading"]), "-stats", *args, "in.png"]
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=900)
        outs = sorted(pathlib.Path(d).glob("*.ktx2" if ktx2 else "*.basis"))
        assert r.returncode == 0 and len(outs) == 1, r.stdout[-2000:] + r.stderr[-2000:]
        if all_psnrs:
            def grab(label):
                mm = re.search(r"^" + label + r" Avg:\s+Max:\s*\S+\s+Mean:\s*\S+\s+RMS:\s*\S+\s+PSNR:\s*([0-9.]+)", r.stdout, re.M)
                return float(mm.group(1)) if mm else None
            return np.fromfile(outs[0], np.uint8), {"rgb": grab("RGB"), "rgba": grab("RGBA"), "bc7_rgba": grab("BC7 RGBA")}
        m = re.search(r"^RGBA Avg:\s+Max:\s*\S+\s+Mean:\s*\S+\s+RMS:\s*\S+\s+PSNR:\s*([0-9.]+)", r.stdout, re.M)
        return np.fromfile(outs[0], np.uint8), (float(m.group(1)) if m else None)


def file_record(data, psnr, ktx2):
    kv = (helpers.ktx2_file_key_values if ktx2 else helpers.basis_file_key_values)(data)
    return {"size": int(data.size), "sha256": sha(data), "key_values": [[k, bytes(v).hex()] for k, v in kv], "tool_psnr_rgba_slice0": psnr}


def uastc_grid(out):
    """the 90 cUASTC_LDR_4x4 rows: the file the tool writes for -quality / -effort, the three PSNRs it prints, and the low-level settings the pair maps to
    (from the reference's own set_format_mode_and_quality_effort through the harness)"""
    tf = helpers.REF_DIR / "test_files"
    grid = out.setdefault("codec_grid_uastc", {})
    import os
    # the table's own rows (file, quality, effort -> ktx2 size, RGB / RGBA / BC7 RGBA PSNR), read from the reference's generated header
    rows = {}
    for m in re.finditer(r'\{ "(\w+)\.png", basist::basis_tex_format::cUASTC_LDR_4x4, (\d+), (\d+), false, (\d+), ([0-9.]+)f, ([0-9.]+)f, ([0-9.]+)f \}',
                         (helpers.REF_DIR / "basisu_tool_test_codecs.inl").read_text()):
        rows[f"{m.group(1)}/q{m.group(2)}/e{m.group(3)}"] = {"size": int(m.group(4)), "rgb": float(m.group(5)), "rgba": float(m.group(6)), "bc7_rgba": float(m.group(7))}
    assert len(rows) == 90, len(rows)
    out["reference_table_codec_grid_uastc"] = rows
    out["codec_grid_uastc_tool_threads"] = os.cpu_count()      # the post-pass takes min(4, pool threads) strips (comp.cpp:2078)
    assert os.cpu_count() >= 4
    for name in CODEC_FILES:
        for q in QUALITIES:
            for e in EFFORTS:
                _, _, flags, rdo, lam = helpers.ref_quality_effort(True, q, e)
                data, psnrs = run_tool(tf / f"{name}.png", "-ktx2_no_zstandard", "-uastc", "-quality", str(q), "-effort", str(e), ktx2=True, multithreaded=True, all_psnrs=True)
                rec = file_record(data, psnrs["rgba"], True)
                rec.update(tool_psnr_rgb=psnrs["rgb"], tool_psnr_bc7_rgba=psnrs["bc7_rgba"], uastc_level=flags, rdo=bool(rdo),
                           rdo_lambda_f32_hex=np.float32(lam).tobytes().hex(), rdo_jobs=4)
                grid[f"{name}/q{q}/e{e}"] = rec
        print("uastc grid", name, flush=True)
        OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")


def main():
    if sys.argv[1:] == ["uastc_grid"]:
        out = json.loads(OUT.read_text())
        uastc_grid(out)
        OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
        return
    tf = helpers.REF_DIR / "test_files"
    extra = {n: helpers.load_png(tf / f"{n}.png") for n in EXTRA}
    np.savez_compressed(NPZ, **extra)
    out = json.loads(OUT.read_text()) if OUT.exists() else {}
    out["reference_table_extra"] = {n: dict(zip(("etc1s_q1_size", "etc1s_q1_psnr", "uastc_psnr", "etc1s_q128_size", "etc1s_q128_psnr"), v)) for n, v in EXTRA_TABLE.items()}
    out.setdefault("table", {})
    names = [f"kodim{k:02d}" for k in range(1, 25)] + list(EXTRA)
    for name in names:
        t0 = time.time()
        png = tf / f"{name}.png"
        rec = out["table"].get(name, {})
        img = helpers.load_png(png)
        rec["width"], rec["height"], rec["rgba_sha256"] = int(img.shape[1]), int(img.shape[0]), sha(img)
        # column 1 of the table: ETC1S quality 1 with the library defaults the -test mode encodes with (comp level 2, linear metrics)
        rec["etc1s_q1_table"] = file_record(*run_tool(png, "-etc1s", "-q", "1", "-comp_level", "2", "-linear"), False)
        if name in EXTRA:
            # the other two columns + the command line's defaults, as gen_golden_kodak.py records them for kodim01..24
            rec["etc1s_q128_table"] = file_record(*run_tool(png, "-etc1s", "-q", "128", "-comp_level", "2", "-linear"), False)
            rec["etc1s_q128"] = file_record(*run_tool(png, "-etc1s", "-q", "128", "-comp_level", "1"), False)
            rec["uastc_l0_file"] = file_record(*run_tool(png, "-uastc", "-uastc_level", "0", "-linear"), False)   # -test: no level bits, no cFlagSRGB
            rec["uastc_l2_file"] = file_record(*run_tool(png, "-uastc"), False)
            blocks = helpers.to_pixel_blocks(img)
            packed = helpers.ref_encode_uastc(blocks, 2)
            rec["n_blocks"] = int(blocks.shape[0])
            rec["uastc_l2"], rec["uastc_l0"] = sha(packed), sha(helpers.ref_encode_uastc(blocks, 0))
            for jobs in (1, 4):
                if blocks.shape[0] >= jobs:
                    rec[f"uastc_l2_rdo1_jobs{jobs}"] = sha(helpers.ref_uastc_rdo(packed, blocks, 2, 0 if jobs == 1 else jobs, lam=1.0))
        out["table"][name] = rec
        print(name, rec["etc1s_q1_table"]["size"], rec["etc1s_q1_table"]["tool_psnr_rgba_slice0"], f"{time.time() - t0:.1f}s", flush=True)
        OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    # the multi-threaded tool writes the same file for the largest of these inputs (413,862 blocks would be needed to reach the partitioned build)
    a, _ = run_tool(tf / "wikipedia.png", "-etc1s", "-q", "128", "-comp_level", "1", multithreaded=True)
    assert sha(a) == out["table"]["wikipedia"]["etc1s_q128"]["sha256"], "the tool's default configuration differs from -no_multithreading on wikipedia.png"
    out["multithreaded_tool_checked_on"] = "wikipedia.png -etc1s -q 128 -comp_level 1"
    # ---- the ETC1S half of the codec grid
    grid = out.setdefault("codec_grid_etc1s", {})
    for name in CODEC_FILES:
        for q in QUALITIES:
            for e in EFFORTS:
                ql, lvl = low_level(q, e)
                data, psnr = run_tool(tf / f"{name}.png", "-etc1s", "-q", str(ql), "-comp_level", str(lvl), ktx2=True)
                grid[f"{name}/q{q}/e{e}"] = dict(file_record(data, psnr, True), etc1s_quality=ql, comp_level=lvl)
        print("grid", name, flush=True)
        OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    uastc_grid(out)
    OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
