// pmx_common.h — shared host/device helpers of libpmesh_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/pmesh_amd.h"

// Timing experiments that compute WRONG results (atomics or weights compiled out, transforms skipped) exist
// only in a build made with -DPMX_EXPERIMENT (scripts/build_variant.sh); a default build cannot contain them:
// naming one of their switches without it is a compile error, and pmx_build_flags() lets a caller see what a
// library was built with (bench.py refuses to print a line for an experiment build).
#ifndef PMX_EXPERIMENT
#if defined(PMX_EXP_NOATOM) || defined(PMX_EXP_NOWEIGHT) || defined(PMX_EXP_NOPASS) || defined(PMX_EXP_BINFLOOR) || defined(PMX_EXP_NODEPOSIT32) || defined(PMX_EXP_LEANBIN)
#error "PMX_EXP_* switches produce wrong results: they need -DPMX_EXPERIMENT as well"
#endif
#endif

namespace pmx {

void set_error(const char *fmt, ...);

#define PMX_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            pmx::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                           hipGetErrorString(_e));                                       \
            return PMX_EHIP;                                                             \
        }                                                                                \
    } while (0)

#define PMX_REQUIRE(cond, code, msg)                                                     \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            pmx::set_error("%s:%d: %s (%s)", __FILE__, __LINE__, msg, #cond);            \
            return code;                                                                 \
        }                                                                                \
    } while (0)

// Device-side copy of a pmx_vec (passed by value as a kernel argument).
struct DVec {
    const char *data;
    int64_t stride0, stride1;
    int32_t elsize;
    __device__ __forceinline__ double get(int64_t i, int c) const
    {
        const char *p = data + i * stride0 + c * stride1;
        return elsize == 8 ? *(const double *)p : (double)*(const float *)p;
    }
    __device__ __forceinline__ void set(int64_t i, int c, double v) const
    {
        char *p = const_cast<char *>(data) + i * stride0 + c * stride1;
        if (elsize == 8) *(double *)p = v;
        else *(float *)p = (float)v;
    }
};

// position loads with the element size known at compile time (PE = 4 / 8) or read from the vector (PE = 0):
// the tile kernels choose once per launch (see tile_gather) instead of once per component
// [r6] Workgroups are handed to the 8 XCDs in turn (blockIdx mod 8), each with an L2 of its own.  Where neighbouring work
// items share HBM lines (column passes over lines off 128-byte boundaries, the face lines of the readout's regions),
// xcd_tile gives every XCD a contiguous range of the n items instead — workgroup b is the (b / 8)-th of XCD b mod 8 —
// so that neighbours in the array are neighbours in time on ONE L2.  A bijection of [0, n) for any n < 2^32.
__device__ __forceinline__ int64_t xcd_tile(int64_t b, int64_t n)
{
    const uint32_t ub = (uint32_t)b, un = (uint32_t)n;
    const uint32_t x = ub & 7u, i = ub >> 3, q = un >> 3, r = un & 7u;
    return (int64_t)(x * q + (x < r ? x : r) + i);
}

template <int PE> __device__ __forceinline__ double pos_get(const DVec &pos, int64_t i, int c)
{
    if (PE == 0) return pos.get(i, c);
    const char *q = pos.data + i * pos.stride0 + c * pos.stride1;
    return PE == 8 ? *(const double *)q : (double)*(const float *)q;
}

inline DVec dvec(const pmx_vec *v)
{
    DVec d;
    if (v && v->data) {
        d.data = (const char *)v->data;
        d.stride0 = v->stride0;
        d.stride1 = v->stride1;
        d.elsize = v->elsize;
    } else {
        d.data = nullptr;
        d.stride0 = d.stride1 = 0;
        d.elsize = 8;
    }
    return d;
}

inline bool vec_ok(const pmx_vec *v) { return v && v->data && (v->elsize == 4 || v->elsize == 8); }

// Grid sizing for HBM-bound streaming kernels: enough workgroups to fill
// 256 CUs several times over, grid-stride for the rest.
inline unsigned grid_for(int64_t n, int block, int64_t cap = 256 * 32)
{
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

}  // namespace pmx
