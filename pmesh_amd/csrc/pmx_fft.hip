// pmx_fft.hip — local FFT stages on rocFFT and the pack/unpack kernels of the
// slab transpose.
//
// Replaces pfft.Plan / plan.execute (pmesh/pm.py:1406-1441, 689, 1017).  PFFT =
// serial FFTW per rank + MPI all-to-all transposes; here the local stages are
// rocFFT plans (batched, strided, in-place capable, with the reference's
// forward normalisation 1/prod(Nmesh) of pm.py:692 folded into the plan's
// scale factor so no separate pass over the mesh is needed) and the global
// transpose is an RCCL all-to-all issued by the host layer between
// pmx_slab_pack (before) / pmx_slab_unpack (after the reverse exchange).
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <mutex>
#include <vector>

#include "pmx_common.h"

struct pmx_fft {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
    size_t work_bytes = 0;
    hipStream_t last_stream = nullptr;
    bool stream_set = false;
};

namespace pmx {
static std::once_flag g_rocfft_once;
// rocFFT plan creation (kernel generation + caches) is not safe to run concurrently from
// several host threads: serialise it.  Execution is lock-free.
static std::mutex g_plan_mutex;

#define PMX_FFT_CHECK(expr)                                                              \
    do {                                                                                 \
        rocfft_status _s = (expr);                                                       \
        if (_s != rocfft_status_success) {                                               \
            pmx::set_error("%s:%d: %s -> rocfft status %d", __FILE__, __LINE__, #expr,   \
                           (int)_s);                                                     \
            return PMX_EFFT;                                                             \
        }                                                                                \
    } while (0)
}  // namespace pmx

using namespace pmx;

extern "C" int pmx_fft_create(pmx_fft **out, int32_t kind, int32_t elsize, int32_t ndim,
                              const int64_t *n, const int64_t *istride, int64_t idist,
                              const int64_t *ostride, int64_t odist, int64_t batch, double scale,
                              int32_t inplace)
{
    PMX_REQUIRE(out != nullptr, PMX_EINVAL, "plan pointer is NULL");
    PMX_REQUIRE(ndim >= 1 && ndim <= 3, PMX_EINVAL, "ndim must be 1..3");
    PMX_REQUIRE(elsize == 4 || elsize == 8, PMX_EINVAL, "elsize must be 4 or 8");
    PMX_REQUIRE(kind >= PMX_FFT_R2C && kind <= PMX_FFT_C2C_BWD, PMX_EINVAL, "bad transform kind");
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
    std::lock_guard<std::mutex> plan_lock(g_plan_mutex);

    rocfft_transform_type tt;
    rocfft_array_type it, ot;
    switch (kind) {
    case PMX_FFT_R2C: tt = rocfft_transform_type_real_forward; it = rocfft_array_type_real; ot = rocfft_array_type_hermitian_interleaved; break;
    case PMX_FFT_C2R: tt = rocfft_transform_type_real_inverse; it = rocfft_array_type_hermitian_interleaved; ot = rocfft_array_type_real; break;
    case PMX_FFT_C2C_FWD: tt = rocfft_transform_type_complex_forward; it = ot = rocfft_array_type_complex_interleaved; break;
    default: tt = rocfft_transform_type_complex_inverse; it = ot = rocfft_array_type_complex_interleaved; break;
    }
    // Measured on this ROCm (scripts/bigfft_probe.py): the in-place 3-d R2C/C2R of a 2048^3 mesh
    // (4.3e9 complex elements) returns wrong numbers without an error, 2048x2048x1024 (2.1e9) is
    // right.  Refuse spans of 2^32 elements or more rather than hand back a wrong field; the own
    // kernels (pmx_rowfft / pmx_colfft, 64-bit offsets throughout) cover the power-of-two meshes.
    {
        double span_i = 0, span_o = 0;
        for (int d = 0; d < ndim; d++) {
            span_i += (double)(n[d] - 1) * (double)istride[d];
            span_o += (double)(n[d] - 1) * (double)ostride[d];
        }
        span_i += (double)(batch - 1) * (double)idist;
        span_o += (double)(batch - 1) * (double)odist;
        const double lim = 4294967296.0 * (kind <= PMX_FFT_C2R ? 2.0 : 1.0);   // real side counts reals
        bool real_in = kind == PMX_FFT_R2C, real_out = kind == PMX_FFT_C2R;
        PMX_REQUIRE(span_i < (real_in ? lim : 4294967296.0) && span_o < (real_out ? lim : 4294967296.0),
                    PMX_EUNSUPPORTED, "rocFFT transform spanning 2^32 or more complex elements");
    }
    // rocFFT wants lengths/strides fastest axis first; the ABI is C order
    size_t len[3], is[3], os[3];
    for (int d = 0; d < ndim; d++) {
        len[d] = (size_t)n[ndim - 1 - d];
        is[d] = (size_t)istride[ndim - 1 - d];
        os[d] = (size_t)ostride[ndim - 1 - d];
    }
    rocfft_plan_description desc = nullptr;
    PMX_FFT_CHECK(rocfft_plan_description_create(&desc));
    PMX_FFT_CHECK(rocfft_plan_description_set_data_layout(desc, it, ot, nullptr, nullptr, ndim, is,
                                                          (size_t)idist, ndim, os, (size_t)odist));
    if (scale != 1.0) PMX_FFT_CHECK(rocfft_plan_description_set_scale_factor(desc, scale));
    pmx_fft *p = new pmx_fft();
    rocfft_status s = rocfft_plan_create(&p->plan, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace,
                                         tt, elsize == 8 ? rocfft_precision_double : rocfft_precision_single,
                                         ndim, len, (size_t)batch, desc);
    rocfft_plan_description_destroy(desc);
    if (s != rocfft_status_success) {
        delete p;
        set_error("rocfft_plan_create failed with status %d", (int)s);
        return PMX_EFFT;
    }
    PMX_FFT_CHECK(rocfft_execution_info_create(&p->info));
    PMX_FFT_CHECK(rocfft_plan_get_work_buffer_size(p->plan, &p->work_bytes));
    if (p->work_bytes) {
        PMX_HIP_CHECK(hipMalloc(&p->work, p->work_bytes));
        PMX_FFT_CHECK(rocfft_execution_info_set_work_buffer(p->info, p->work, p->work_bytes));
    }
    *out = p;
    return PMX_OK;
}

extern "C" int pmx_fft_execute(pmx_fft *p, void *in, void *outb, void *stream)
{
    PMX_REQUIRE(p && p->plan, PMX_EINVAL, "plan is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (!p->stream_set || st != p->last_stream) {
        PMX_FFT_CHECK(rocfft_execution_info_set_stream(p->info, st));
        p->last_stream = st;
        p->stream_set = true;
    }
    void *ib[1] = {in};
    void *ob[1] = {outb};
    PMX_FFT_CHECK(rocfft_execute(p->plan, ib, (outb && outb != in) ? ob : nullptr, p->info));
    return PMX_OK;
}

extern "C" int pmx_fft_destroy(pmx_fft *p)
{
    if (!p) return PMX_OK;
    if (p->info) rocfft_execution_info_destroy(p->info);
    if (p->plan) rocfft_plan_destroy(p->plan);
    if (p->work) (void)hipFree(p->work);
    delete p;
    return PMX_OK;
}

// ---- slab transpose pack/unpack --------------------------------------------
namespace pmx {

struct Offs { int64_t v[PMX_MAXRANKS + 1]; int32_t n; };

// src (n0, n1, n2) -> block r holds (n0, n1[r]..n1[r+1], n2), blocks concatenated
template <typename E>
__global__ void __launch_bounds__(256) slab_pack_kernel(const E *src, E *dst, int64_t n0, int64_t n1,
                                                        int64_t n2, Offs o, bool inverse)
{
    const int64_t total = n0 * n1 * n2;
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total;
         q += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = q % n2, r = q / n2;
        int64_t j = r % n1, i = r / n1;
        int part = 0;
        while (j >= o.v[part + 1]) part++;
        int64_t w = o.v[part + 1] - o.v[part];
        int64_t p = o.v[part] * n0 * n2 + (i * w + (j - o.v[part])) * n2 + k;
        if (!inverse) dst[p] = src[q];
        else dst[q] = src[p];
    }
}

struct alignas(8) E8 { float a, b; };
struct alignas(16) E16 { double a, b; };

static int load_offs(Offs &o, const int64_t *h, int nparts)
{
    PMX_REQUIRE(nparts >= 1 && nparts <= PMX_MAXRANKS, PMX_EUNSUPPORTED, "more than 64 parts");
    o.n = nparts;
    for (int i = 0; i <= nparts; i++) o.v[i] = h[i];
    return PMX_OK;
}

}  // namespace pmx

static int slab_pack_impl(const void *src, void *dst, int64_t n0, int64_t n1, int64_t n2,
                          const int64_t *n1_offsets, int32_t nparts, int32_t elbytes, void *stream,
                          bool inverse)
{
    PMX_REQUIRE(elbytes == 8 || elbytes == 16, PMX_EINVAL, "elbytes must be 8 or 16");
    Offs o;
    int rc = load_offs(o, n1_offsets, nparts);
    if (rc) return rc;
    PMX_REQUIRE(o.v[nparts] == n1, PMX_EINVAL, "n1_offsets must end at n1");
    int64_t total = n0 * n1 * n2;
    if (total == 0) return PMX_OK;
    hipStream_t st = (hipStream_t)stream;
    if (elbytes == 16)
        slab_pack_kernel<E16><<<grid_for(total, 256), 256, 0, st>>>((const E16 *)src, (E16 *)dst, n0, n1, n2, o, inverse);
    else
        slab_pack_kernel<E8><<<grid_for(total, 256), 256, 0, st>>>((const E8 *)src, (E8 *)dst, n0, n1, n2, o, inverse);
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

extern "C" int pmx_slab_pack(const void *src, void *dst, int64_t n0, int64_t n1, int64_t n2,
                             const int64_t *n1_offsets, int32_t nparts, int32_t elbytes, void *stream)
{
    return slab_pack_impl(src, dst, n0, n1, n2, n1_offsets, nparts, elbytes, stream, false);
}

extern "C" int pmx_slab_unpack(const void *src, void *dst, int64_t n0, int64_t n1, int64_t n2,
                               const int64_t *n1_offsets, int32_t nparts, int32_t elbytes, void *stream)
{
    return slab_pack_impl(src, dst, n0, n1, n2, n1_offsets, nparts, elbytes, stream, true);
}
