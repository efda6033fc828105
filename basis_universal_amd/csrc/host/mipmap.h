// mipmap.h -- host half of mip generation: what basis_compressor::generate_mipmaps -> image_resample -> Resampler
// (encoder/basisu_comp.cpp:2146-2230, basisu_enc.cpp:1022-1180, basisu_resampler.cpp, basisu_resample_filters.cpp) decide on the host
// before any pixel is touched -- the separable filter's contributor lists (which source samples feed a destination sample, with what
// float weight), the order of the two passes, the sRGB tables -- so that the device applies them (mipmap_kernels.hip) with the same
// float operations in the same order and writes the reference's bytes.
//
// The weights come out of float / double libm arithmetic (sin, cos, exp, log, sqrt, powf), evaluated here exactly as written there;
// this file is compiled by the same compiler family with no contraction, and the GPU never re-derives them.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace bu {
namespace mip {

// ---- reconstruction filters (basisu_resample_filters.cpp): f(t), support
inline double sinc(double x) {
    x = x * 3.14159265358979323846;
    if ((x < 0.01f) && (x > -0.01f)) return 1.0f + x * x * (-1.0f / 6.0f + x * x * 1.0f / 120.0f);
    return std::sin(x) / x;
}
inline float clean(double t) { return std::fabs(t) < .0000125f ? 0.0f : (float)t; }
inline double blackman_window(double x) { return 0.42659071f + 0.49656062f * std::cos(3.14159265358979323846 * x) + 0.07684867f * std::cos(2.0f * 3.14159265358979323846 * x); }
inline double bessel_i0(double x) {
    double half = 0.5 * x, sum = 1.0, term = 1.0, ds = 1.0;
    for (int k = 1; ds > sum * 1E-16; k++) { term = term * (half / k); ds = term * term; sum = sum + ds; }
    return sum;
}
inline float f_box(float t) { return (t >= -0.5f && t < 0.5f) ? 1.0f : 0.0f; }
inline float f_tent(float t) { if (t < 0.0f) t = -t; return t < 1.0f ? 1.0f - t : 0.0f; }
inline float f_bell(float t) {
    if (t < 0.0f) t = -t;
    if (t < .5f) return .75f - (t * t);
    if (t < 1.5f) { t = t - 1.5f; return .5f * (t * t); }
    return 0.0f;
}
inline float cubic_bc(float t, const float B, const float C) {
    const float tt = t * t;
    if (t < 0.0f) t = -t;
    if (t < 1.0f) { t = ((12.0f - 9.0f * B - 6.0f * C) * (t * tt)) + ((-18.0f + 12.0f * B + 6.0f * C) * tt) + (6.0f - 2.0f * B); return t / 6.0f; }
    if (t < 2.0f) { t = ((-1.0f * B - 6.0f * C) * (t * tt)) + ((6.0f * B + 30.0f * C) * tt) + ((-12.0f * B - 48.0f * C) * t) + (8.0f * B + 24.0f * C); return t / 6.0f; }
    return 0.0f;
}
inline float f_mitchell(float t) { return cubic_bc(t, 1.0f / 3.0f, 1.0f / 3.0f); }
inline float f_catmullrom(float t) { return cubic_bc(t, 0.0f, .5f); }
template <int N> inline float f_lanczos(float t) { if (t < 0.0f) t = -t; return t < (float)N ? clean(sinc(t) * sinc(t / (float)N)) : 0.0f; }
inline float f_blackman(float t) { if (t < 0.0f) t = -t; return t < 3.0f ? clean(sinc(t) * blackman_window(t / 3.0f)) : 0.0f; }
inline float f_kaiser(float t) {
    if (t < 0.0f) t = -t;
    if (t < 3) {
        const float att = 40.0f;
        const float alpha = (float)(std::exp(std::log((double)0.58417 * (att - 20.96)) * 0.4) + 0.07886 * (att - 20.96));
        const double ratio = (double)t / 3.0;
        return (float)clean(sinc(t) * (bessel_i0(alpha * std::sqrt(1 - ratio * ratio)) / bessel_i0(alpha)));
    }
    return 0.0f;
}
struct filter { const char* name; float (*f)(float); float support; };
inline const filter* find_filter(const char* name) {
    static const filter table[] = {{"box", f_box, 0.5f}, {"tent", f_tent, 1.0f}, {"bell", f_bell, 1.5f}, {"mitchell", f_mitchell, 2.0f}, {"blackman", f_blackman, 3.0f},
                                   {"lanczos3", f_lanczos<3>, 3.0f}, {"lanczos4", f_lanczos<4>, 4.0f}, {"lanczos6", f_lanczos<6>, 6.0f}, {"lanczos12", f_lanczos<12>, 12.0f},
                                   {"kaiser", f_kaiser, 3.0f}, {"catmullrom", f_catmullrom, 2.0f}};
    for (const filter& e : table) if (!std::strcmp(e.name, name)) return &e;
    return nullptr;
}

// ---- contributor lists of one axis (Resampler::make_clist, basisu_resampler.cpp:62-341), CSR
struct contributors {
    std::vector<uint32_t> first;    // dst + 1 offsets
    std::vector<uint16_t> pixel;    // source sample (after the boundary rule)
    std::vector<float> weight;
    uint32_t ops() const { return (uint32_t)pixel.size(); }
};
inline int wrap_index(int j, int n, bool wrap) {  // Resampler::reflect for BOUNDARY_WRAP / BOUNDARY_CLAMP
    if (j < 0) { if (!wrap) return 0; const int m = (-j) % n; return m ? n - m : 0; }
    if (j >= n) return wrap ? j % n : n - 1;
    return j;
}
inline bool make_contributors(contributors& out, int src_n, int dst_n, bool wrap, const filter& flt, float filter_scale) {
    out.first.assign(1, 0); out.pixel.clear(); out.weight.clear();
    const float oo_filter_scale = 1.0f / filter_scale, nudge = 0.5f;
    const float xscale = dst_n / (float)src_n;
    const bool minify = xscale < 1.0f;
    const float half_width = minify ? (flt.support / xscale) * filter_scale : flt.support * filter_scale;
    for (int i = 0; i < dst_n; i++) {
        float center = ((float)i + nudge) / xscale;
        center -= nudge;
        center += 0.0f;
        const int left = (int)(float)std::floor(center - half_width), right = (int)(float)std::ceil(center + half_width);
        auto w_of = [&](int j) { return minify ? flt.f((center - (float)j) * xscale * oo_filter_scale) : flt.f((center - (float)j) * oo_filter_scale); };
        float total = 0;
        for (int j = left; j <= right; j++) total += w_of(j);
        const float norm = (float)(1.0f / total);
        total = 0;
        int max_k = -1;
        float max_w = -1e+20f;
        const size_t base = out.pixel.size();
        for (int j = left; j <= right; j++) {
            const float w = w_of(j) * norm;
            if (w == 0.0f) continue;
            out.pixel.push_back((uint16_t)wrap_index(j, src_n, wrap));
            out.weight.push_back(w);
            total += w;
            if (w > max_w) { max_w = w; max_k = (int)(out.pixel.size() - base) - 1; }
        }
        if (max_k == -1) return false;
        if (total != 1.0f) out.weight[base + max_k] += 1.0f - total;
        out.first.push_back((uint32_t)out.pixel.size());
    }
    return true;
}

// ---- one resampling step = image_resample(src, dst, srgb, filter, scale, wrapping, first_comp 0, num_comps) as data for the device
struct plan {
    contributors x, y;
    bool x_after_y = false;            // Resampler's m_delay_x_resample: the cheaper order by its operation count (resampler.cpp:773-789)
    float srgb_to_linear[256];
    uint8_t linear_to_srgb[8192];
};
inline bool make_plan(plan& p, uint32_t src_w, uint32_t src_h, uint32_t dst_w, uint32_t dst_h, bool srgb, const char* filter_name, float filter_scale, bool wrap) {
    const filter* flt = find_filter(filter_name);
    if (!flt || !src_w || !src_h || !dst_w || !dst_h || src_w > 16384 || src_h > 16384) return false;
    if (!make_contributors(p.x, (int)src_w, (int)dst_w, wrap, *flt, filter_scale) || !make_contributors(p.y, (int)src_h, (int)dst_h, wrap, *flt, filter_scale)) return false;
    const int x_ops = (int)p.x.ops(), y_ops = (int)p.y.ops();
    const int xy_ops = x_ops * (int)src_h + (4 * y_ops * (int)dst_w) / 3, yx_ops = (4 * y_ops * (int)src_w) / 3 + x_ops * (int)dst_h;
    p.x_after_y = (xy_ops > yx_ops) || (xy_ops == yx_ops && src_w < dst_w);
    auto sat = [](float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); };
    for (int i = 0; i < 256; i++) {  // basisu_enc.cpp:395-411, 1061-1075
        const float s = (float)i * (1.0f / 255.0f);
        p.srgb_to_linear[i] = srgb ? (s < .04045f ? sat(s * (1.0f / 12.92f)) : sat(powf((s + .055f) * (1.0f / 1.055f), 2.4f))) : s;
    }
    for (int i = 0; i < 8192; i++) {
        const float l = (float)i * (1.0f / 8191);
        const float v = l < .0031308f ? sat(l * 12.92f) : sat(1.055f * powf(l, 1.0f / 2.4f) - .055f);
        const int q = (int)(255.0f * v + .5f);
        p.linear_to_srgb[i] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
    return true;
}

// mip chain dimensions (comp.cpp:2153-2160, 2203-2204)
inline std::vector<std::pair<uint32_t, uint32_t>> level_sizes(uint32_t w, uint32_t h, uint32_t smallest_dimension) {
    std::vector<std::pair<uint32_t, uint32_t>> out;
    uint32_t lw = w, lh = h, levels = 1;
    while (std::max(lw, lh) > smallest_dimension) { lw = std::max(lw >> 1, 1u); lh = std::max(lh >> 1, 1u); levels++; }
    for (uint32_t l = 1; l < levels; l++) out.emplace_back(std::max(1u, w >> l), std::max(1u, h >> l));
    return out;
}

}  // namespace mip
}  // namespace bu
