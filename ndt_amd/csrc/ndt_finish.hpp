// ndt_finish.hpp -- one primary node's colour -> its pixel (the end of get_pixel_color, ndt.c:488-568): shared by the per-frame
// kernel k_finish_pixels (ndt_frame.hip) and the streaming frame kernel (ndt_stream.hpp), which finishes a pixel the moment its
// ray tree is resolved.
#pragma once
#include "ndt_kernels.hpp"

// (COH: inside the frame kernel the node's colour may have been written by another wavefront of the same launch:
// agent-scope loads, like every hand-off there)
template <bool COH> __device__ __forceinline__ double fin_ld(const double *p)
{
    if (COH)
        return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long *>(const_cast<double *>(p)),
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}
template <bool COH> __device__ __forceinline__ int fin_ldi(const int *p)
{
    if (COH) return __hip_atomic_load(const_cast<int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// get_pixel_color's adaptive loop (ndt.c:488-568) replayed on the one deterministic sample:
// with samples == 1 the reference re-traces the identical ray k times, k decided by the
// running-mean test below; the result is (c+...+c)/k and the k-fold ray count.
// Returns k x (rays of the node's tree): what the reference's trace_kd counter gains by this pixel.
template <bool COH>
__device__ __forceinline__ unsigned long long finish_pixel(const double *blob, const SceneDesc &sd, const Workspace &ws, const RenderGeom &rg,
                                                           int N_, long long g, double *rgba, double *depth_out)
{
    long long out_idx = g;                          // list mode: one colour per sample
    if (!rg.samples) {
        const int tile = (int)(g >> 6), lane = (int)(g & 63);
        const int px = (tile % rg.tiles_x) * 8 + (lane & 7);
        const int py = (tile / rg.tiles_x) * 8 + (lane >> 3);
        out_idx = (long long)py * rg.width + px;    // dbl_image_set_pixel, image.c:126
    }
    const double l[4] = { fin_ld<COH>(ws.clr + 0 * ws.cap + g), fin_ld<COH>(ws.clr + 1 * ws.cap + g), fin_ld<COH>(ws.clr + 2 * ws.cap + g),
                          fin_ldi<COH>(ws.hit_obj + g) >= 0 ? 1.0 : blob[sd.off_cam + 4 * N_ + 7] };
    double t[4] = { 0.0, 0.0, 0.0, 0.0 };
    const double max_diff = 1.0 / 256.0;
    double clr_diff = 256;
    int samples = 0;
    // Every sample is the same colour l, so the reference's
    //     clr_diff = max_c |t_c/(i-1) - (t_c+l_c)/i|        (t = l+l+...+l, i terms)
    // is max_c(l_c)/(i(i-1)) up to rounding (relative error < 4 i^2 ulp: a difference of two
    // quotients of an i-term running sum).  The six divisions are only spent when that
    // estimate lies inside the error band around 1/256; otherwise the loop-exit decision
    // is already certain and identical to the exact one.
    const double gb0 = (fabs(l[1]) > fabs(l[2])) ? fabs(l[1]) : fabs(l[2]);
    const double lmax = (fabs(l[0]) > gb0) ? fabs(l[0]) : gb0;
    const bool finite = lmax <= 1.0e300;          // false for inf / nan: always take the exact path
    for (int i = 0; i < 1 || (!rg.raw_samples && i < 10000 && clr_diff > max_diff); ++i) {
        if (i > 1) {
            const double ii = (double)i * (double)(i - 1);
            const double est = lmax / ii;
            const double band = 1.0e-15 * (8.0 * (double)i * (double)i) + 1.0e-12;
            if (finite && est > max_diff * (1.0 + band)) {
                clr_diff = est;             // certainly still above the threshold: keep sampling
            } else if (finite && est < max_diff * (1.0 - band)) {
                clr_diff = est;             // certainly converged: the loop ends here
            } else {
                const double dr = fabs(t[0] / (i - 1) - (t[0] + l[0]) / i);
                const double dg = fabs(t[1] / (i - 1) - (t[1] + l[1]) / i);
                const double db = fabs(t[2] / (i - 1) - (t[2] + l[2]) / i);
                const double gb = (dg > db) ? dg : db;      // MAX, image.h:31
                clr_diff = (dr > gb) ? dr : gb;
            }
        }
        t[0] += l[0]; t[1] += l[1]; t[2] += l[2]; t[3] += l[3];
        samples += 1;
    }
    double *out = rgba + out_idx * 4;
    out[0] = t[0] / samples;
    out[1] = t[1] / samples;
    out[2] = t[2] / samples;
    out[3] = t[3] / samples;
    if (depth_out) depth_out[out_idx] = fin_ld<COH>(ws.depth + g);       // ndt.c:753-756
    return (unsigned long long)samples * (unsigned long long)fin_ldi<COH>(ws.count + g);
}
