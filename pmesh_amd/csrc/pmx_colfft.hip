// pmx_colfft.hip — batched, strided ("column") complex FFT of power-of-two length,
// in place, with whole columns resident in LDS.
//
// Why: in a 3-d transform of an (N0, N1, N2c) array the passes along axes 1 and 0 walk
// the array with a stride of one row / one plane.  rocFFT's kernels for that shape
// (`..._sbcc_...`, 4 columns = 64-byte rows per workgroup) reach 2.9 TB/s and over-fetch
// ~30 % at 512^3 (profiles/, PMC FETCH_SIZE); they are 2/3 of the r2c/c2r time, which
// in turn is 44 % of the PM cycle.  Here one workgroup owns W adjacent columns
// (one 128-byte row segment per row: 8 double or 16 float columns), loads them once, runs a Stockham
// radix-8/4/2 FFT entirely in LDS, and stores them once: one read + one write of the
// array per pass, nothing else.  The transform of the PM cycle is completed by rocFFT's
// unit-stride R2C/C2R along the contiguous axis (which already runs near the copy rate).
//
// Array view: (A, N, B) complex, C order; the FFT runs along the middle axis (stride B),
// batched over A (stride N*B) and B.  Axis-1 pass of (N0, N1, N2c): A=N0, N=N1, B=N2c;
// axis-0 pass: A=1, N=N0, B=N1*N2c.  Unnormalised in both directions; `scale`
// multiplies the result (the forward transform of the cycle carries 1/prod(Nmesh),
// pmesh/pm.py:692).  An optional transfer function (same closed forms as
// pmx_transfer.hip, SIMPLE variant) can be applied while loading, which folds
// ComplexField.apply into the first pass of c2r.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "pmx_common.h"
#include "pmx_binplan.h"     // tile geometry of the paint whose halos the forward row pass can gather

// This file compiles as one unit (PMX_COLFFT_PART undefined or 0: scripts/build_variant.sh, the resource test) or as two,
// built side by side by the Makefile: the C ABI with the double-precision kernels (1) and the float kernels
// (2: pmx_colfft_f4.hip) — 80 s of compilation otherwise, the longest step of a clean build.
#ifndef PMX_COLFFT_PART
#define PMX_COLFFT_PART 0
#endif
#if PMX_COLFFT_PART == 0
#define PMX_DISPATCH static
#else
#define PMX_DISPATCH
#endif

namespace pmx {

template <typename T> struct cpx { T x, y; };

template <typename T> __device__ __forceinline__ cpx<T> cmul(cpx<T> a, cpx<T> b)
{
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T> __device__ __forceinline__ cpx<T> cadd(cpx<T> a, cpx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ cpx<T> csub(cpx<T> a, cpx<T> b) { return {a.x - b.x, a.y - b.y}; }
// multiply by -i (forward) or +i (inverse)
template <typename T, bool INV> __device__ __forceinline__ cpx<T> rot90(cpx<T> a)
{
    if (INV) return {-a.y, a.x};
    return {a.y, -a.x};
}

template <typename T, bool INV> __device__ __forceinline__ void fft2(cpx<T> *v)
{
    cpx<T> a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

template <typename T, bool INV> __device__ __forceinline__ void fft4(cpx<T> *v)
{
    cpx<T> a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
    cpx<T> c = cadd(v[1], v[3]), d = rot90<T, INV>(csub(v[1], v[3]));
    v[0] = cadd(a, c);
    v[2] = csub(a, c);
    v[1] = cadd(b, d);
    v[3] = csub(b, d);
}

// in-register DFT of 8 points, natural order in and out
template <typename T, bool INV> __device__ __forceinline__ void fft8(cpx<T> *v)
{
    const T h = (T)0.70710678118654752440;
    // radix-2 stage: pairs (k, k+4)
    cpx<T> a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    cpx<T> a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    cpx<T> a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    cpx<T> a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    // twiddles w8^k on the odd half: w8 = exp(-+ i pi/4)
    {
        cpx<T> t = a5;   // * w8^1
        if (INV) a5 = {h * (t.x - t.y), h * (t.x + t.y)};
        else a5 = {h * (t.x + t.y), h * (t.y - t.x)};
        a6 = rot90<T, INV>(a6);   // * w8^2 = -+ i
        t = a7;          // * w8^3
        if (INV) a7 = {-h * (t.x + t.y), h * (t.x - t.y)};
        else a7 = {h * (t.y - t.x), -h * (t.x + t.y)};
    }
    // two DFT-4: even outputs from a0..a3, odd outputs from a4..a7
    cpx<T> e[4] = {a0, a1, a2, a3}, o[4] = {a4, a5, a6, a7};
    fft4<T, INV>(e);
    fft4<T, INV>(o);
    v[0] = e[0]; v[2] = e[1]; v[4] = e[2]; v[6] = e[3];
    v[1] = o[0]; v[3] = o[1]; v[5] = o[2]; v[7] = o[3];
}

// 3 points: w = exp(-+ 2 pi i / 3) = -1/2 -+ i sqrt(3)/2
template <typename T, bool INV> __device__ __forceinline__ void fft3(cpx<T> *v)
{
    const T s = (T)0.86602540378443864676;
    cpx<T> t1 = cadd(v[1], v[2]);
    cpx<T> t2 = {v[0].x - (T)0.5 * t1.x, v[0].y - (T)0.5 * t1.y};
    cpx<T> t3 = {s * (v[1].x - v[2].x), s * (v[1].y - v[2].y)};
    cpx<T> r = rot90<T, INV>(t3);          // -+ i t3
    v[0] = cadd(v[0], t1);
    v[1] = cadd(t2, r);
    v[2] = csub(t2, r);
}

// 5 points
template <typename T, bool INV> __device__ __forceinline__ void fft5(cpx<T> *v)
{
    const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;   // cos(2 pi/5), cos(4 pi/5)
    const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;    // sin(2 pi/5), sin(4 pi/5)
    cpx<T> t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]), t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    cpx<T> a1 = {v[0].x + c1 * t1.x + c2 * t2.x, v[0].y + c1 * t1.y + c2 * t2.y};
    cpx<T> a2 = {v[0].x + c2 * t1.x + c1 * t2.x, v[0].y + c2 * t1.y + c1 * t2.y};
    cpx<T> b1 = rot90<T, INV>(cpx<T>{s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y});
    cpx<T> b2 = rot90<T, INV>(cpx<T>{s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y});
    v[0] = cadd(v[0], cadd(t1, t2));
    v[1] = cadd(a1, b1);
    v[4] = csub(a1, b1);
    v[2] = cadd(a2, b2);
    v[3] = csub(a2, b2);
}

template <typename T, bool INV, int R> __device__ __forceinline__ void fftR(cpx<T> *v)
{
    if (R == 8) fft8<T, INV>(v);
    else if (R == 4) fft4<T, INV>(v);
    else if (R == 3) fft3<T, INV>(v);
    else if (R == 5) fft5<T, INV>(v);
    else fft2<T, INV>(v);
}

// Length codes of the kernel templates: LC < 16 is N = 2^LC, LC = 16 + k is N = 3 * 2^k
// (192, 384, 768, 1536 — the 3 * 2^k meshes of production runs).
// and LC = 32 + k is N = 5 * 2^k (320, 640, 1280).
template <int LC> struct Len { static constexpr int N = LC < 16 ? (1 << LC) : (LC < 32 ? (3 << (LC - 16)) : (5 << (LC - 32))); };
static inline int length_code(int64_t n)
{
    if (n <= 0) return -1;
    if ((n & (n - 1)) == 0) { int l = 0; while ((1ll << l) < n) l++; return l; }
    if (n % 3 == 0) {
        int64_t q = n / 3;
        if ((q & (q - 1)) == 0) { int k = 0; while ((1ll << k) < q) k++; return 16 + k; }
    }
    if (n % 5 == 0) {
        int64_t q = n / 5;
        if ((q & (q - 1)) == 0) { int k = 0; while ((1ll << k) < q) k++; return 32 + k; }
    }
    return -1;
}

// Addressing of element (a, n, b): a*sa + (n >> sh)*shi + (n & mask)*sn + b.  The plain
// (A, N, B) array is sh = 31, mask = ~0, sa = N*B, sn = B (or padded strides: the one-rank
// complex layout pads its plane stride, see fft.py).  The "split" layout of the slab transpose
// cuts axis N into N/nl ranges of nl = 1 << sh lines, one contiguous (A, nl, B) block per
// range (= per destination rank): sa = nl*B, shi = A*nl*B, mask = nl-1.
// Column remap (cw > 0): column b of the batch is column (b / cw) * cpitch + b % cw of the
// array — the B = n1 * cw columns of a chunk [coff, coff + cw) of the last axis of an
// (N, n1, cpitch) array (coff is folded into the base pointer).  This is how the chunks of a
// pipelined slab transpose are gathered from / scattered into the standard layout.
struct ColAddr {
    int64_t sa, shi, sn;   // sn: stride between successive lines n (B when dense)
    int64_t cw, cpitch;
    int32_t sh, mask;
};

__device__ __forceinline__ int64_t col_offset(const ColAddr &a, int64_t b)
{
    if (a.cw <= 0) return b;
    int64_t q = b / a.cw;
    return q * a.cpitch + (b - q * a.cw);
}

// offset of line n of a column.  With n = tj + c, tj < TPC, c a multiple of TPC and TPC, 1 << sh powers
// of two it splits into line_offset(tj) + line_offset(c) — a per-thread part computed once and a part
// that is the same for the whole workgroup (scalar unit); the kernels of the power-of-two lengths
// address their lines that way instead of four 64-bit multiply-adds per element.
__device__ __forceinline__ int64_t line_offset(const ColAddr &a, int n)
{
    return (int64_t)(n >> a.sh) * a.shi + (int64_t)(n & a.mask) * a.sn;
}

// a workgroup-uniform offset, pinned to scalar registers where it is used: in the loop of the pipelined
// kernel the compiler otherwise hoists the RPT line offsets, parks them in VECTOR registers across the
// passes and spills them
template <bool PIN> __device__ __forceinline__ int64_t uniform_offset(int64_t off)
{
    if (PIN) asm volatile("" : "+s"(off));
    return off;
}

struct ColGeom {
    int64_t A, B;          // outer and inner batch extents
    int32_t N, logN;
    ColAddr in, out;
    double scale;
    // optional fused transfer (APPLY): global index bookkeeping of the (N0, n1, N2c) block
    pmx_transfer t;
    int32_t n1, n2;        // B = n1 * n2 for the axis-0 pass
    int64_t start[3], nmesh[3];
    double dw[3], nl[3];
    // [r4] grad_kind 1 (force_transfer, nbody.py:162-171): D(k_d) = (8 sin w - sin 2w) / (6 C), w = k_d C, C = L / N — a
    // function of the global index along ONE axis: a table of nmesh[grad_dir] doubles made on the host.  Along axes 1
    // and 2 the factor belongs to the COLUMN (column_k: one cached load per tile and thread, nothing per element);
    // along axis 0 it would be a load per element — in the round-trip kernels, which sit at their register limit, that
    // spills (measured: 60-116 bytes in every float variant) — so that direction keeps its stand-alone kernel.
    // nullptr: D = k_d.
    const double *dtab = nullptr;
    // [r6] tiles in XCD order (xcd_tile below): set by the launchers
    int32_t xcd = 0;
};

// LDS layout: one row of the tile = W columns = 128 bytes = half a bank row.  Rows 2m and
// 2m+1 share bank row m; WHICH half a row takes is XOR-swizzled with bits 3, 6, 9 of the
// row number, so that the row sets a wave touches in every Stockham pass (consecutive rows
// when reading, rows 8 or 64 apart when writing) always split evenly over the two halves:
// all passes are bank-conflict free without padding.
template <typename T, int RB = 128, bool ROT = true> __device__ __forceinline__ int lds_index(int row, int col)
{
    constexpr int W = RB / (int)sizeof(cpx<T>);
    // ROT = false: the column kernels' tiles, without the rotation of the column slot (see RowBase below)
    const int slot = ROT ? (col + row) & (W - 1) : col;
    if (RB == 256) {
        // a tile row is a whole bank row: any set of rows is conflict free; the column slot is
        // still rotated by the row number for the transposing accesses
        return row * W + slot;
    }
    // parity of bits 0, 3, 6, 9 in two folds
    int x = row ^ (row >> 3);
    x ^= x >> 6;
    // the column slot is rotated by the row number: a wave that walks along a column
    // (the transposing load/store of the row kernel) then also spreads over all banks
    return ((row & ~1) | (x & 1)) * W + slot;
}

// The same index in two steps, for rows of the form `row + c` with c a compile-time constant whose set
// bits are clear in `row` (row + c == row | c: no carries).  Every term of lds_index is then additive or
// XOR-linear in the two parts, so the per-thread part is computed once (RowBase) and every access costs
// a constant that the compiler folds into the offset field of the ds_read / ds_write:
//   (row + c) & ~1 = (row & ~1) + (c & ~1),   half(row + c) = half(row) ^ half(c),
//   rotation (col + row + c) & (W - 1): independent of c when c is a multiple of W.
// The power-of-two lengths address every line this way (rows tj + u * TPC, butterfly legs j + r * N/R,
// outputs base + r * Ns); before, the swizzle was recomputed for every element — 8 integer operations
// against the 6 floating-point ones of the butterfly itself.
// ROT = false: without the rotation of the column slot.  The rotation only matters to accesses that walk
// along a column of the tile with consecutive lanes (the transposing load / store of the row kernel); the
// column kernels never do — their lanes always cover whole 128-byte tile rows, and permuting the slots
// inside a row changes nothing for the banks — so their tiles are laid out without it and the column slot
// costs no instruction at all.
template <typename T, int RB> struct RowBase { int e[2]; int cr; };

template <typename T, int RB, bool ROT = true> __device__ __forceinline__ RowBase<T, RB> row_base(int row, int col)
{
    constexpr int W = RB / (int)sizeof(cpx<T>);
    RowBase<T, RB> b;
    b.cr = ROT ? col + row : col;
    if (RB == 256) {
        b.e[0] = b.e[1] = row * W;
    } else {
        const int half = (row ^ (row >> 3) ^ (row >> 6) ^ (row >> 9)) & 1;
        b.e[0] = ((row & ~1) + half) * W;
        b.e[1] = ((row & ~1) + (half ^ 1)) * W;
    }
    if (!ROT) { b.e[0] += col; b.e[1] += col; }
    return b;
}

template <typename T, int RB, bool ROT = true> __device__ __forceinline__ int lds_at(const RowBase<T, RB> &b, int c)
{
    constexpr int W = RB / (int)sizeof(cpx<T>);
    const int rot = ROT ? (b.cr + c) & (W - 1) : 0;
    if (RB == 256) return b.e[0] + c * W + rot;
    const int hc = (c ^ (c >> 3) ^ (c >> 6) ^ (c >> 9)) & 1;
    return b.e[hc] + (c & ~1) * W + rot;
}

// One pass of the Stockham autosort FFT over the LDS-resident tile.  Lane mapping: the W
// columns of a row are W consecutive lanes (col fastest), so a wave works on 64/W
// butterflies of all W columns at once.
// HALFTW: `tw` holds only the first N/2 entries of the table (w[m + N/2] = -w[m]); used where the
// full table would not fit beside the tile (N = 2048 in double).
template <typename T, bool INV, int R, int RB = 128, bool HALFTW = false, bool ROT = true>
__device__ __forceinline__ void stockham_pass(cpx<T> *buf, const cpx<T> *tw, int N, int Ns, int tpc /*threads per column*/,
                                              int col, int tj, int twstride = 1)
{
    // every thread handles (N/R)/tpc butterflies of its column
    const int nb = N / R;
    const int per = (nb + tpc - 1) / tpc;   // radix 3 with tpc = N/8: 8/3 -> 3 trips, the last one partial
    cpx<T> v[4][8];   // up to 4 butterflies of radix <= 8 per thread (tpc = N/8, R = 2 -> 4)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (q >= per) break;
        int j = tj + q * tpc;
        if (j >= nb) break;
        int k = j % Ns;
#pragma unroll
        for (int r = 0; r < R; r++) {
            cpx<T> x = buf[lds_index<T, RB, ROT>(j + r * nb, col)];
            if (r > 0 && Ns > 1) {
                // twiddle exp(-+ 2 pi i r k / (Ns R)) from the length-N table
                int m = r * k * (N / (Ns * R));
                cpx<T> w;
                if (HALFTW) {
                    w = tw[m & (N / 2 - 1)];
                    if (m & (N / 2)) { w.x = -w.x; w.y = -w.y; }
                } else {
                    w = tw[m * twstride];
                }
                if (INV) w.y = -w.y;
                x = cmul(x, w);
            }
            v[q][r] = x;
        }
        fftR<T, INV, R>(v[q]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (q >= per) break;
        int j = tj + q * tpc;
        if (j >= nb) break;
        int k = j % Ns;
        int base = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; r++) buf[lds_index<T, RB, ROT>(base + r * Ns, col)] = v[q][r];
    }
    __syncthreads();
}

// The pass for power-of-two lengths: everything but the lane's position is a compile-time constant.
// tb = row_base(tj, col) of the calling kernel: the legs of butterfly j = tj + q * TPC are the rows
// tj + (q * TPC + r * N/R) (tj < TPC <= N/R: disjoint bits).  The arithmetic is that of stockham_pass,
// operation for operation: the results are bit-identical.
// LOP: an operation on every element as it is loaded, given its row n (the round-trip kernel's scale and transfer
// function ride on the loads of the first inverse pass: the rows a thread loads there are the rows it would have
// rewritten in a sweep of its own, with a barrier and an LDS round trip in between)
struct NoLoadOp {
    static constexpr bool active = false;
    template <typename V> __device__ __forceinline__ V operator()(int, V x) const { return x; }
};
// IO (register passes, [r5]): bit 0 — the legs of this pass are already in `io` (io[q + r per] = row tj + (q + r per) TPC:
// for the FIRST pass of a transform, NS = 1, these are exactly the RPT = R per lines tj + u TPC a column kernel's thread
// has loaded from memory); bit 1 — the results stay in `io` in the same numbering (for the LAST pass, NS = N / R, the
// rows a Stockham pass writes, j + r NS, are again the thread's own lines tj + u TPC).  Same operations in the same
// order as through LDS: the same bits; what is saved is a sweep of the tile through LDS each way and its barriers.
template <typename T, bool INV, int R, int RB, bool HALFTW, int N, int NS, int TPC, int TWS, bool ROT, typename LOP = NoLoadOp, int IO = 0>
__device__ __forceinline__ void stockham_pass_p2(cpx<T> *buf, const cpx<T> *tw, const RowBase<T, RB> &tb, int col, int tj,
                                                 const LOP &lop = LOP(), cpx<T> *io = nullptr)
{
    static_assert(!(IO & 1) || NS == 1, "register legs: the first pass only");
    static_assert(!(IO & 2) || NS * R == N, "register results: the last pass only");
    constexpr int nb = N / R;
    constexpr int per = nb / TPC;
    static_assert((N & (N - 1)) == 0 && (TPC & (TPC - 1)) == 0 && nb % TPC == 0 && per >= 1 && per <= 4,
                  "power-of-two pass: butterflies per thread");
    cpx<T> v[per][R];
#pragma unroll
    for (int q = 0; q < per; q++) {
        const int j = tj + q * TPC;
        const int k = j & (NS - 1);
#pragma unroll
        for (int r = 0; r < R; r++) {
            cpx<T> x = (IO & 1) ? io[q + r * per] : buf[lds_at<T, RB, ROT>(tb, q * TPC + r * nb)];
            if (LOP::active) x = lop(tj + q * TPC + r * nb, x);
            if (r > 0 && NS > 1) {
                const int m = r * k * (N / (NS * R));
                cpx<T> w;
                if (HALFTW) {
                    w = tw[(m & (N / 2 - 1)) * TWS];
                    if (m & (N / 2)) { w.x = -w.x; w.y = -w.y; }
                } else {
                    w = tw[m * TWS];
                }
                if (INV) w.y = -w.y;
                x = cmul(x, w);
            }
            v[q][r] = x;
        }
        fftR<T, INV, R>(v[q]);
    }
    if (IO & 2) {
        // (rows j + r NS with NS = nb = per TPC: line q + r per of this thread; nothing is written: no barrier — the
        // next pass that writes the tile waits for this one's reads itself)
#pragma unroll
        for (int q = 0; q < per; q++)
#pragma unroll
            for (int r = 0; r < R; r++) io[q + r * per] = v[q][r];
        return;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < per; q++) {
        const int j = tj + q * TPC;
        const int k = j & (NS - 1);
        // outputs: rows (j - k) * R + k + r * NS — k in the low bits, r above them, j / NS on top
        const RowBase<T, RB> wb = row_base<T, RB, ROT>((j - k) * R + k, col);
#pragma unroll
        for (int r = 0; r < R; r++) buf[lds_at<T, RB, ROT>(wb, r * NS)] = v[q][r];
    }
    __syncthreads();
}

// Fused transfer (SIMPLE forms).  The wavenumbers along axes 1 and 2 depend only on the
// column, so they are computed once per tile and thread; per element only axis 0 remains.
// k12sq = k1^2 + k2^2; dcol: the gradient factor D of a gradient along axis 1 or 2 (k1 / k2, or the table's entry) —
// all a column contributes to its elements' transfer function
struct ColK { double k12sq, dcol; };

__device__ __forceinline__ double kcoord(const ColGeom &g, int d, int64_t i)
{
    // (mesh sides are below 2^31: 32-bit integers convert to double in one instruction, 64-bit ones in four)
    const int32_t gi = (int32_t)i + (int32_t)g.start[d];
    double wi = (double)gi;
    if (gi >= (int32_t)g.nmesh[d] / 2) wi -= (double)(int32_t)g.nmesh[d];
    wi *= g.dw[d];
    return wi * g.nl[d];
}

__device__ __forceinline__ ColK column_k(const ColGeom &g, int64_t b)
{
    const uint32_t ub = (uint32_t)b, un2 = (uint32_t)g.n2;
    const uint32_t i1 = ub / un2, i2 = ub - i1 * un2;
    ColK c;
    const double k1 = kcoord(g, 1, i1), k2 = kcoord(g, 2, i2);
    c.k12sq = k1 * k1 + k2 * k2;
    c.dcol = g.t.grad_dir == 1 ? k1 : k2;
    if (g.dtab && g.t.grad_dir > 0)
        c.dcol = g.dtab[g.t.grad_dir == 1 ? (int32_t)i1 + (int32_t)g.start[1] : (int32_t)i2 + (int32_t)g.start[2]];
    return c;
}

// FORM 1: the transfer is i k_d / k^2 times an amplitude (laplace_pow == -1, grad_dir >= 0: every force component of
// a PM step) — the same operations as the general form below in the same order, without its questions: per element
// the general form asks three wave-uniform ones, which the compiler turns into branches (and register moves at
// their joins) in every one of the RPT unrolled copies.  The kernels ask once, around the loop.
template <typename T, int FORM = 0>
__device__ __forceinline__ cpx<T> apply_simple(const ColGeom &g, int64_t i0, const ColK &c, cpx<T> v)
{
    if (FORM == 1) {
        const double k0 = kcoord(g, 0, i0);
        const double k2 = k0 * k0 + c.k12sq;
        const double qq = (k2 == 0) ? 1.0 : k2;
        double re = g.t.amplitude;
        re *= 1.0 / qq;
        const double D = g.t.grad_dir == 0 ? k0 : c.dcol;     // (a select, no branch)
        const double im = re * D;
        re = 0;
        const double ar = v.x, ai = v.y;
        return {(T)(re * ar - im * ai), (T)(re * ai + im * ar)};
    }
    const double k0 = kcoord(g, 0, i0);
    const double k2 = k0 * k0 + c.k12sq;
    double re = g.t.amplitude, im = 0;
    if (g.t.laplace_pow) {
        double qq = (k2 == 0) ? 1.0 : k2;
        if (g.t.laplace_pow == -1) re *= 1.0 / qq;
        else if (g.t.laplace_pow == 1) re *= qq;
    }
    if (g.t.grad_dir >= 0) {
        const double D = g.t.grad_dir == 0 ? k0 : c.dcol;
        im = re * D;
        re = 0;
    }
    double ar = v.x, ai = v.y;
    return {(T)(re * ar - im * ai), (T)(re * ai + im * ar)};
}

// radices per log2(N): products of 8/4/2, largest first
template <int LOGN> struct Radices;
template <> struct Radices<6>  { static constexpr int n = 2; static constexpr int r[4] = {8, 8, 1, 1}; };
template <> struct Radices<7>  { static constexpr int n = 3; static constexpr int r[4] = {8, 4, 4, 1}; };
template <> struct Radices<8>  { static constexpr int n = 3; static constexpr int r[4] = {8, 8, 4, 1}; };
template <> struct Radices<9>  { static constexpr int n = 3; static constexpr int r[4] = {8, 8, 8, 1}; };
template <> struct Radices<10> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 4, 4}; };
template <> struct Radices<11> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 8, 4}; };
template <> struct Radices<22> { static constexpr int n = 3; static constexpr int r[4] = {8, 8, 3, 1}; };   // 192
template <> struct Radices<23> { static constexpr int n = 4; static constexpr int r[4] = {8, 4, 4, 3}; };   // 384
template <> struct Radices<24> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 4, 3}; };   // 768
template <> struct Radices<25> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 8, 3}; };   // 1536
template <> struct Radices<38> { static constexpr int n = 3; static constexpr int r[4] = {8, 8, 5, 1}; };   // 320
template <> struct Radices<39> { static constexpr int n = 4; static constexpr int r[4] = {8, 4, 4, 5}; };   // 640
template <> struct Radices<40> { static constexpr int n = 4; static constexpr int r[4] = {8, 8, 4, 5}; };   // 1280

// [r6] Workgroups are handed to the 8 XCDs in turn (blockIdx mod 8), each with an L2 of its own.  A pass over lines that
// do not start on 128-byte boundaries — the dense blocks of the distributed transforms' transposes: 257 or 65 modes to
// a line — has neighbouring tiles share every HBM line their column groups straddle; with tile = blockIdx the two
// halves are fetched by two XCDs.  xcd_tile (pmx_common.h) gives every XCD a contiguous range of the tiles instead
// (workgroup b is the (b / 8)-th of XCD b mod 8): neighbours in the array are neighbours in time on one L2.  Measured
// (scripts/r06/col_xcd_ab.sh, profiles/r06_colxcd/): passes over dense lines 3.99 -> 4.72-4.76 TB/s at N = 512 in double
// (257 modes to a line), 3.6 -> 4.6-4.8 in float, the round trip at N = 1024 in double 3.74 -> 3.98 on padded lines, 2.65 ->
// 3.85 on dense ones — but the y pass of the ONE-rank transform, padded lines, inside its L3-sized blocks of planes
// LOSES (512^3: r2c 0.83 -> 0.90 ms, float 0.48 -> 0.53).  So: 1 (default) = the round trip always, the plain passes
// where a line does not start on a 128-byte boundary; 2 = every pass (the measurement build); 0 = none.
#ifndef PMX_COL_XCD
#define PMX_COL_XCD 1
#endif

// column kernel: lengths from this one on walk their tiles with a grid-stride loop
#ifndef PMX_COL_STRIDE_FROM
#define PMX_COL_STRIDE_FROM 1000000
#endif
#ifndef PMX_RPT_D1024
#define PMX_RPT_D1024 8
#endif
// lines of a column per thread: 8 for double; 16 for float, whose 8-byte elements would
// otherwise keep only half the bytes in flight per thread (measured 3.1 vs 4.6 TB/s per pass)
#ifndef PMX_RPT_D
#define PMX_RPT_D 8
#endif
#ifndef PMX_RPT_F
#define PMX_RPT_F 16
#endif
template <typename T, int LOGN> struct Rpt { static constexpr int value = LOGN >= 16 ? 8 : PMX_RPT_D; };
// (3 * 2^k lengths: 8 in both precisions — the radix-3 pass of a thread with 16 lines would need
// 6 butterflies in registers)
template <int LOGN> struct Rpt<float, LOGN> { static constexpr int value = LOGN >= 16 ? 8 : PMX_RPT_F; };
// N = 1024 in double: the tile takes 147 KB of LDS, one workgroup per CU.  1024 threads with 8
// lines each: r2c / c2r 12.7 / 12.7 ms at 1024^3 against 13.4 / 13.5 ms with 512 threads of 16
// lines (which was the faster one while the tile loop kept 156 VGPRs alive); 64-byte tile rows
// with two workgroups per CU measured the same as this (13.0 / 12.7 ms).
template <> struct Rpt<double, 10> { static constexpr int value = PMX_RPT_D1024; };
// N = 2048: 64-byte row segments (8 float / 4 double columns per tile) keep the tile inside the
// 160 KB of LDS; 16 lines per thread in float, 8 in double (1024 threads each; 16 lines / 512
// threads in double measured 3 % slower at 2048^3)
#ifndef PMX_RPT_D2048
#define PMX_RPT_D2048 8
#endif
template <> struct Rpt<double, 11> { static constexpr int value = PMX_RPT_D2048; };

// the twiddle table shares the LDS with the tile; where the pair would exceed ~150 KB only its
// first half is kept (see stockham_pass)
template <typename T, int LOGN, int RB> struct HalfTw {
    static constexpr size_t full = (size_t)(Len<LOGN>::N * (RB / (int)sizeof(cpx<T>)) + Len<LOGN>::N) * sizeof(cpx<T>);
    // (also where the half table lets two workgroups share a CU: 2 x 80 KB would fill the LDS to the byte)
    static constexpr bool value = (LOGN < 16) && (full > 150 * 1024 || (full > 76 * 1024 && full <= 80 * 1024));
};

// tiles of which only one fits a CU beside its twiddles: the column kernel pipelines them (see there)
#ifndef PMX_COL_PIPE
#define PMX_COL_PIPE 1
#endif
template <typename T, int LOGN, int RB> struct ColPipe {
    static constexpr size_t bytes = (size_t)(Len<LOGN>::N * (RB / (int)sizeof(cpx<T>))
                                             + (HalfTw<T, LOGN, RB>::value ? Len<LOGN>::N / 2 : Len<LOGN>::N)) * sizeof(cpx<T>);
    // (N = 2048 in float, 16 lines per thread, spills under the prefetch: left as it was)
    static constexpr bool value = PMX_COL_PIPE && bytes > 80 * 1024 && !(sizeof(T) == 4 && LOGN == 11);
};

// The round-trip kernel (colfft_round_kernel) has the same persistent form behind PMX_ROUND_PIPE, off by default: its
// two transforms and the double-precision transfer arithmetic leave no room for the 32 registers of the prefetched
// tile within 128 VGPRs (1024 threads), the compiler spills the prefetched lines themselves, and 1024^3 in double
// went from 6.9 to 8.6 ms per launch; with 512 threads of 16 lines (256 VGPRs) 7.3 ms.
#ifndef PMX_ROUND_PIPE
#define PMX_ROUND_PIPE 0
#endif
// [r4] PMX_ROUND_PIPE2: the same persistent, prefetching form for the tiles of which TWO fit a CU (N = 512 in double:
// 74 KB, 90 VGPRs — the 32 registers of the prefetched tile stay inside the 128 that four waves per SIMD leave): two
// workgroups per CU, each loading its next tile while the two transforms of the current one run.
#ifndef PMX_ROUND_PIPE2
#define PMX_ROUND_PIPE2 0
#endif
#ifndef PMX_ROUND_REGS
#define PMX_ROUND_REGS 1
#endif
// (where the compiler keeps both transforms' registers and the transfer arithmetic inside 128 VGPRs: lengths up to 512
// in double; the others spill 16-140 bytes per lane in this form and keep the LDS round trips)
#ifndef PMX_COL_REGS
#define PMX_COL_REGS 1
#endif
template <typename T, int LOGN, bool APPLY, bool REMAP> struct ColRegs { static constexpr bool value = PMX_COL_REGS != 0; };
// (PMX_ROUND_REGS_F4 / PMX_ROUND_REGS_MAXLOG: measurement builds of the forms that spill — scripts/r06/round_regs_ab.sh)
#ifndef PMX_ROUND_REGS_F4
#define PMX_ROUND_REGS_F4 0
#endif
#ifndef PMX_ROUND_REGS_MAXLOG
#define PMX_ROUND_REGS_MAXLOG 9
#endif
template <typename T, int LOGN> struct RoundRegs { static constexpr bool value = PMX_ROUND_REGS != 0 && (sizeof(T) == 8 || PMX_ROUND_REGS_F4) && LOGN <= PMX_ROUND_REGS_MAXLOG; };
template <typename T, int LOGN, int RB> struct RoundPipe2 {
    static constexpr bool value = PMX_ROUND_PIPE2 && LOGN < 16 && !ColPipe<T, LOGN, RB>::value && ColPipe<T, LOGN, RB>::bytes > 53 * 1024;
};
template <typename T, int LOGN, int RB> struct RoundPipe {
    static constexpr bool value = (PMX_ROUND_PIPE && ColPipe<T, LOGN, RB>::value) || RoundPipe2<T, LOGN, RB>::value;
};

// the Stockham passes of an N-point transform over the LDS-resident tile (compile-time radices)
// REGIO: the first pass takes its legs from `io`, the last leaves its results there (stockham_pass_p2: IO)
template <typename T, int LOGN, bool INV, int RB, bool HT, int TPC, int TWS, bool ROT, int I, int NS, typename LOP = NoLoadOp, bool REGIO = false>
__device__ __forceinline__ void run_passes_p2(cpx<T> *buf, const cpx<T> *tw, const RowBase<T, RB> &tb, int col, int tj,
                                              const LOP &lop = LOP(), cpx<T> *io = nullptr)
{
    using Rd = Radices<LOGN>;
    if constexpr (I < Rd::n) {
        constexpr int IO = REGIO ? ((I == 0 ? 1 : 0) | (I == Rd::n - 1 ? 2 : 0)) : 0;
        // (the load operation belongs to the first pass only)
        if constexpr (I == 0) stockham_pass_p2<T, INV, Rd::r[I], RB, HT, Len<LOGN>::N, NS, TPC, TWS, ROT, LOP, IO>(buf, tw, tb, col, tj, lop, io);
        else stockham_pass_p2<T, INV, Rd::r[I], RB, HT, Len<LOGN>::N, NS, TPC, TWS, ROT, NoLoadOp, IO>(buf, tw, tb, col, tj, NoLoadOp(), io);
        run_passes_p2<T, LOGN, INV, RB, HT, TPC, TWS, ROT, I + 1, NS * Rd::r[I], NoLoadOp, REGIO>(buf, tw, tb, col, tj, NoLoadOp(), io);
    }
}

template <typename T, int LOGN, bool INV, int RB, bool HT>
__device__ __forceinline__ void run_passes(cpx<T> *buf, const cpx<T> *tw, int col, int tj)
{
    constexpr int N = Len<LOGN>::N;
    constexpr int TPC = N / Rpt<T, LOGN>::value;
    if constexpr (LOGN < 16) {
        const RowBase<T, RB> tb = row_base<T, RB, false>(tj, col);      // column kernels: tiles without the rotation
        run_passes_p2<T, LOGN, INV, RB, HT, TPC, 1, false, 0, 1>(buf, tw, tb, col, tj);
        return;
    }
    int Ns = 1;
    using Rd = Radices<LOGN>;
    if (Rd::r[0] == 8) stockham_pass<T, INV, 8, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
    Ns *= Rd::r[0];
    if (Rd::n > 1) {
        if (Rd::r[1] == 8) stockham_pass<T, INV, 8, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        else if (Rd::r[1] == 4) stockham_pass<T, INV, 4, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        else if (Rd::r[1] == 3) stockham_pass<T, INV, 3, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        else if (Rd::r[1] == 5) stockham_pass<T, INV, 5, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        Ns *= Rd::r[1];
    }
    if (Rd::n > 2) {
        if (Rd::r[2] == 8) stockham_pass<T, INV, 8, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        else if (Rd::r[2] == 4) stockham_pass<T, INV, 4, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        else if (Rd::r[2] == 3) stockham_pass<T, INV, 3, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        else if (Rd::r[2] == 5) stockham_pass<T, INV, 5, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        Ns *= Rd::r[2];
    }
    if (Rd::n > 3) {
        if (Rd::r[3] == 8) stockham_pass<T, INV, 8, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        else if (Rd::r[3] == 4) stockham_pass<T, INV, 4, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        else if (Rd::r[3] == 3) stockham_pass<T, INV, 3, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        else if (Rd::r[3] == 5) stockham_pass<T, INV, 5, RB, HT, false>(buf, tw, N, Ns, TPC, col, tj);
        Ns *= Rd::r[3];
    }
}

// REMAP: the columns go through col_offset (chunks of a pipelined transpose).  A template
// parameter because the per-lane column offsets cost the plain passes 34 VGPRs (156 instead of
// 122 at N = 512: one workgroup per CU instead of two, 541 instead of 476 us per pass).
template <typename T, int LOGN, bool INV, bool APPLY, int RB, bool REMAP>
__global__ void __launch_bounds__((Len<LOGN>::N / Rpt<T, LOGN>::value * (RB / (int)sizeof(cpx<T>))))
colfft_kernel(ColGeom g, const cpx<T> *src, cpx<T> *dst, const cpx<T> *twiddle)
{
    constexpr int N = Len<LOGN>::N;
    constexpr int W = RB / (int)sizeof(cpx<T>);    // columns per tile: RB-byte row segments
    constexpr int RPT = Rpt<T, LOGN>::value;       // lines per thread
    constexpr int TPC = N / RPT;                   // threads per column
    constexpr int NT = TPC * W;                    // threads per workgroup
    extern __shared__ __align__(16) unsigned char smem[];
    cpx<T> *buf = reinterpret_cast<cpx<T> *>(smem);
    cpx<T> *tw = buf + N * W;
    constexpr bool HT = HalfTw<T, LOGN, RB>::value;
    const int tid = threadIdx.x;
    for (int n = tid; n < (HT ? N / 2 : N); n += NT) tw[n] = twiddle[n];

    const int64_t tilesB = (g.B + W - 1) / W;
    const int64_t ntiles = g.A * tilesB;
    const int col_ = tid % W, tj_ = tid / W;   // W consecutive lanes = one 128-byte row segment
    // One workgroup per tile (grid = ntiles), and the loop says so: with a grid-stride loop the
    // optimiser keeps everything tile-invariant alive across the FFT passes — 156 / 166 / 212
    // VGPRs for the chunk, fused-transfer and float variants, i.e. one workgroup per CU instead of
    // two (fused pass 680 -> 490 us at 512^3, float passes 343 -> 250 us).
    // PIPE (tiles that leave room for ONE workgroup per CU: N = 1024 / 768 / ... in double): the workgroup is
    // persistent and loads its NEXT tile into registers while the passes of the current one run out of LDS.
    // With a single workgroup on the CU nothing else would overlap the three phases — measured at 1024^3
    // in double, 17 us per tile = 6 us load + 5 us passes + 6 us store, where the memory side alone needs 12.
    // (not the float kernels with the fused transfer: its double-precision arithmetic beside the prefetched tile
    // spills a few registers at N = 1024 / 2048)
    constexpr bool PIPE = ColPipe<T, LOGN, RB>::value && !(APPLY && sizeof(T) == 4);
    constexpr bool ONE_TILE = !PIPE && N < PMX_COL_STRIDE_FROM;
    constexpr bool P2 = LOGN < 16;      // power-of-two length: additive addressing (line_offset, lds_at)
    const int64_t step = ONE_TILE ? ntiles : (int64_t)gridDim.x;
    // (the launcher keeps the tile count below 2^31: a 32-bit division, a dozen scalar instructions
    // instead of the ~150 of the 64-bit one)
    auto place = [&](int64_t tile, int64_t &a, int64_t &b0) __attribute__((always_inline)) {
        const uint32_t ua = (uint32_t)tile / (uint32_t)tilesB;
        a = ua;
        b0 = (int64_t)((uint32_t)tile - ua * (uint32_t)tilesB) * W;
    };
    // load: RPT rows per thread, all loads issued before the first LDS store
    auto load_tile = [&](int64_t tile, cpx<T> (&ld)[RPT]) __attribute__((always_inline)) {
        int tj = tj_, col = col_;
        if (PIPE) asm volatile("" : "+v"(tj), "+v"(col));       // (no hoisting of the addresses: see the loop below)
        int64_t a, b0;
        place(tile, a, b0);
        const bool colok = b0 + col < g.B;
        // plain: one base per tile, lanes add their column; REMAP: a base per lane
        const cpx<T> *ibase = src + a * g.in.sa + (REMAP ? (colok ? col_offset(g.in, b0 + col) : 0) : b0);
        const int lcol = REMAP ? 0 : col;
        const cpx<T> *ithread = ibase + (line_offset(g.in, tj) + lcol);
        int64_t uoff[RPT];
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            ld[u] = cpx<T>{0, 0};
            uoff[u] = P2 ? uniform_offset<PIPE>(line_offset(g.in, u * TPC)) : 0;
        }
        if (colok) {
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                int n = tj + u * TPC;
                if (P2) ld[u] = ithread[uoff[u]];
                else ld[u] = ibase[(int64_t)(n >> g.in.sh) * g.in.shi + (int64_t)(n & g.in.mask) * g.in.sn + lcol];
            }
        }
    };
    cpx<T> ld[RPT];
    auto do_tile = [&](int64_t tile) __attribute__((always_inline)) {
        // (the lane's position is made opaque per trip: everything derived from it — the LDS addresses of
        // every pass — would otherwise be hoisted out of the loop and held in registers across it, 70+
        // VGPRs spilled at N = 1024; recomputing them costs a few dozen instructions per tile)
        int tj = tj_, col = col_;
        if (!ONE_TILE) asm volatile("" : "+v"(tj), "+v"(col));
        const RowBase<T, RB> tb = row_base<T, RB, false>(tj, col);
        int64_t a, b0;
        place(tile, a, b0);
        const bool colok = b0 + col < g.B;
        cpx<T> *obase = dst + a * g.out.sa + (REMAP ? (colok ? col_offset(g.out, b0 + col) : 0) : b0);
        const int lcol = REMAP ? 0 : col;
        ColK ck = {0, 0};
        if (APPLY && colok) ck = column_k(g, b0 + col);
        // [r5] REGS (stockham_pass_p2: IO): the loaded lines are the legs of the first pass, the results of the last
        // pass are the lines to store — the tile goes through LDS 4 times instead of 8, behind 4 barriers instead of
        // 8; the same bits.  (No barrier at the top: the first pass that writes the tile waits for the last reads of the
        // tile before it, and for the twiddles, itself.)
        constexpr bool REGS = ColRegs<T, LOGN, APPLY, REMAP>::value && P2 && !PIPE;
        if constexpr (REGS) {
            cpx<T> *othread = obase + (line_offset(g.out, tj) + lcol);
            load_tile(tile, ld);
            if (APPLY && colok) {
#pragma unroll
                for (int u = 0; u < RPT; u++) ld[u] = apply_simple<T>(g, tj + u * TPC, ck, ld[u]);
            }
            run_passes_p2<T, LOGN, INV, RB, HT, TPC, 1, false, 0, 1, NoLoadOp, true>(buf, tw, tb, col, tj, NoLoadOp(), ld);
            const T sc = (T)g.scale;
            if (colok) {
#pragma unroll
                for (int u = 0; u < RPT; u++) {
                    cpx<T> v = ld[u];
                    v.x *= sc;
                    v.y *= sc;
                    othread[line_offset(g.out, u * TPC)] = v;
                }
            }
            return;
        }
        __syncthreads();
        cpx<T> *othread = obase + (line_offset(g.out, tj) + lcol);
        if (!PIPE) load_tile(tile, ld);
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            int n = tj + u * TPC;
            cpx<T> v = ld[u];
            if (APPLY && colok) v = apply_simple<T>(g, n, ck, v);
            buf[P2 ? lds_at<T, RB, false>(tb, u * TPC) : lds_index<T, RB, false>(n, col)] = v;
        }
        __syncthreads();
        if (PIPE && tile + step < ntiles) load_tile(tile + step, ld);
#if !(defined(PMX_EXPERIMENT) && defined(PMX_EXP_NOPASS))      // (timing experiment, wrong numbers: the kernel's loads and stores without its transform)
        run_passes<T, LOGN, INV, RB, HT>(buf, tw, col, tj);
#endif
        // store
        const T sc = (T)g.scale;
        int64_t uoff[RPT];
#pragma unroll
        for (int u = 0; u < RPT; u++) uoff[u] = P2 ? uniform_offset<PIPE>(line_offset(g.out, u * TPC)) : 0;
        auto store_lines = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < RPT; u++) {
                int n = tj + u * TPC;
                cpx<T> v = buf[P2 ? lds_at<T, RB, false>(tb, u * TPC) : lds_index<T, RB, false>(n, col)];
                v.x *= sc;
                v.y *= sc;
                if (P2) othread[uoff[u]] = v;
                else obase[(int64_t)(n >> g.out.sh) * g.out.shi + (int64_t)(n & g.out.mask) * g.out.sn + lcol] = v;
            }
        };
        if (PIPE) {
            // whole tiles store without a lane mask, i.e. without a branch the wave could skip: only then does
            // the compiler know that RPT stores follow the prefetch loads (see the peeled first tile below).  The
            // ragged last tile of a plane drains its memory operations instead.
            if (b0 + W <= g.B) {
                store_lines();
            } else {
                if (colok) store_lines();
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
            }
        } else if (colok) {
            store_lines();
        }
    };
    int64_t tile = blockIdx.x;
    // (persistent / grid-stride forms: XCD order within every round of gridDim.x tiles)
    if (PMX_COL_XCD && g.xcd) tile = xcd_tile(tile, ONE_TILE ? ntiles : (int64_t)gridDim.x);
    if (PIPE) {
        // The first tile is peeled off the loop: the loop is then only ever entered with "8 loads, then 8 stores"
        // in flight, and the wait for the prefetched lines at its top lets the stores behind them drain on their
        // own (s_waitcnt vmcnt(15..8)).  Entered straight from the prologue's loads, the compiler has to merge the
        // two states into vmcnt(7..0): every tile would wait for the write acknowledgements of the one before.
        if (tile >= ntiles) return;
        load_tile(tile, ld);
        do_tile(tile);
        tile += step;
    }
    for (; tile < ntiles; tile += step) do_tile(tile);
}

// The last pass of r2c and the first pass of c2r run along the same axis: for a caller that goes r2c -> transfer ->
// c2r back to back they are ONE kernel — forward transform of the column, times the forward scale, times the
// transfer function, inverse transform, with the column in LDS throughout: the spectrum is neither written nor read
// (one sweep of the array instead of two; pmx_colfft_roundtrip).  Every value takes exactly the roundings of the
// two separate passes (the scaled mode is rounded to T before the transfer multiplies it), so the result is
// bit-identical to them.  Plain layout, A = 1 (the axis-0 pass of one block), in place.
// (register budget: two workgroups per CU, as the plain passes run — unconstrained, the float kernel of N = 512 took
// 162 VGPRs and ran one: 0.50 ms for the 1.07 GB it moves)
#ifndef PMX_ROUND_WAVES
#define PMX_ROUND_WAVES 4
#endif
#ifndef PMX_ROUND_FUSE_OP
#define PMX_ROUND_FUSE_OP 0       // (measured: 512^3 f8 c2r 1.240 -> 1.230 ms, f4 0.760 -> 0.902, 1024^3 f8 11.97 -> 12.80: the transfer arithmetic in the butterfly registers costs the other lengths their occupancy)
#endif
// what the round-trip kernel does to a mode between its two transforms: times the forward scale, then the transfer
// function (FORM as in apply_simple; APPLY false or a column beyond the block: the scale alone)
template <typename T, bool APPLY, int FORM> struct RoundOp {
    static constexpr bool active = true;
    const ColGeom &g;
    const ColK &ck;
    T sc;
    bool colok;
    __device__ __forceinline__ cpx<T> operator()(int n, cpx<T> v) const
    {
        v.x *= sc;
        v.y *= sc;
        if (APPLY && colok) v = apply_simple<T, FORM>(g, n, ck, v);
        return v;
    }
};

template <typename T, int LOGN, bool APPLY, int RB>
__global__ void __launch_bounds__((Len<LOGN>::N / Rpt<T, LOGN>::value * (RB / (int)sizeof(cpx<T>))),
                                   ((RoundPipe<T, LOGN, RB>::value && !RoundPipe2<T, LOGN, RB>::value) ? 1 : PMX_ROUND_WAVES))
colfft_round_kernel(ColGeom g, cpx<T> *data, const cpx<T> *twiddle)
{
    constexpr int N = Len<LOGN>::N;
    constexpr int W = RB / (int)sizeof(cpx<T>);
    constexpr int RPT = Rpt<T, LOGN>::value;
    constexpr int TPC = N / RPT;
    constexpr int NT = TPC * W;
    extern __shared__ __align__(16) unsigned char smem[];
    cpx<T> *buf = reinterpret_cast<cpx<T> *>(smem);
    cpx<T> *tw = buf + N * W;
    constexpr bool HT = HalfTw<T, LOGN, RB>::value;
    const int tid = threadIdx.x;
    for (int n = tid; n < (HT ? N / 2 : N); n += NT) tw[n] = twiddle[n];
    const int64_t tilesB = (g.B + W - 1) / W;
    const int col_ = tid % W, tj_ = tid / W;
    // PIPE: as in colfft_kernel (see RoundPipe for why it is off)
    constexpr bool PIPE = RoundPipe<T, LOGN, RB>::value;
    constexpr bool P2 = LOGN < 16;
    const int64_t step = PIPE ? (int64_t)gridDim.x : tilesB;
    auto load_tile = [&](int64_t tile, cpx<T> (&ld)[RPT]) __attribute__((always_inline)) {
        int tj = tj_, col = col_;
        if (PIPE) asm volatile("" : "+v"(tj), "+v"(col));
        const int64_t b0 = tile * W;
        const bool colok = b0 + col < g.B;
        const cpx<T> *ithread = data + b0 + ((int64_t)tj * g.in.sn + col);
        int64_t uoff[RPT];
#pragma unroll
        for (int u = 0; u < RPT; u++) {
            ld[u] = cpx<T>{0, 0};
            uoff[u] = uniform_offset<PIPE>((int64_t)(u * TPC) * g.in.sn);
        }
        if (colok) {
#pragma unroll
            for (int u = 0; u < RPT; u++) ld[u] = ithread[uoff[u]];
        }
    };
    cpx<T> ld[RPT];
    auto do_tile = [&](int64_t tile) __attribute__((always_inline)) {
        int tj = tj_, col = col_;
        if (PIPE) asm volatile("" : "+v"(tj), "+v"(col));
        const RowBase<T, RB> tb = row_base<T, RB, false>(tj, col);
        const int64_t b0 = tile * W;
        const bool colok = b0 + col < g.B;
        ColK ck = {0, 0};
        if (APPLY && colok) ck = column_k(g, b0 + col);
        cpx<T> *othread = data + b0 + ((int64_t)tj * g.out.sn + col);
        // [r5] REGS (power-of-two lengths, one tile per workgroup): the lines a thread loads are the legs of its first
        // forward butterflies, what its last forward pass produces are the legs of its first inverse butterflies, and
        // what its last inverse pass produces are the lines it stores (stockham_pass_p2: IO) — the tile goes through
        // LDS 8 times per round trip instead of 13, behind 8 barriers instead of 13; the same bits.
        constexpr bool REGS = RoundRegs<T, LOGN>::value && P2 && !PIPE;
        if constexpr (REGS) {
            const T sc = (T)g.scale;
            const bool force_form = APPLY && !(sizeof(T) == 4 && LOGN == 11) && g.t.laplace_pow == -1 && g.t.grad_dir >= 0;
            __syncthreads();
            load_tile(tile, ld);
            run_passes_p2<T, LOGN, false, RB, HT, TPC, 1, false, 0, 1, NoLoadOp, true>(buf, tw, tb, col, tj, NoLoadOp(), ld);
            // what the forward pass would have stored and the inverse pass loaded: the mode times the forward scale, then
            // the transfer function
            if (force_form) {
                const RoundOp<T, APPLY, 1> op{g, ck, sc, colok};
                // (one line at a time: interleaved, the double-precision transfer arithmetic of all RPT lines beside the
                // tile's registers spills 44-112 bytes per lane at every length but 512 in double)
#pragma unroll
                for (int u = 0; u < RPT; u++) { ld[u] = op(tj + u * TPC, ld[u]); if (APPLY) __builtin_amdgcn_sched_barrier(0); }
            } else {
                const RoundOp<T, APPLY, 0> op{g, ck, sc, colok};
#pragma unroll
                for (int u = 0; u < RPT; u++) { ld[u] = op(tj + u * TPC, ld[u]); if (APPLY) __builtin_amdgcn_sched_barrier(0); }
            }
            run_passes_p2<T, LOGN, true, RB, HT, TPC, 1, false, 0, 1, NoLoadOp, true>(buf, tw, tb, col, tj, NoLoadOp(), ld);
            if (colok) {
#pragma unroll
                for (int u = 0; u < RPT; u++) othread[(int64_t)(u * TPC) * g.out.sn] = ld[u];
            }
            return;
        }
        __syncthreads();
        if (!PIPE) load_tile(tile, ld);
#pragma unroll
        for (int u = 0; u < RPT; u++) buf[P2 ? lds_at<T, RB, false>(tb, u * TPC) : lds_index<T, RB, false>(tj + u * TPC, col)] = ld[u];
        __syncthreads();
        if (PIPE && tile + step < tilesB) load_tile(tile + step, ld);
        run_passes<T, LOGN, false, RB, HT>(buf, tw, col, tj);
        // what the forward pass would have stored and the inverse pass loaded: the mode times the forward scale,
        // then the transfer function
        const T sc = (T)g.scale;
        // (PIPE: one line at a time — unrolled, the double-precision transfer arithmetic of all RPT lines is
        // interleaved and, with the prefetched tile held in registers, spills 90-200 VGPRs)
        // (not N = 2048 in float: the second copy of the loop costs that kernel 18 spilled registers)
        const bool force_form = APPLY && !(sizeof(T) == 4 && LOGN == 11) && g.t.laplace_pow == -1 && g.t.grad_dir >= 0;      // apply_simple<T, 1>
        // [r4] power-of-two lengths: scale and transfer on the loads of the first inverse pass — the rows tj + u TPC this
        // thread would rewrite below are exactly the legs of its first inverse butterflies (N / R = TPC for radix 8, a
        // multiple of it for radix 4), so the values are the same bit for bit, and one LDS round trip with its barrier
        // (of seven per tile) is gone
        constexpr bool FUSE_OP = PMX_ROUND_FUSE_OP && P2 && !PIPE;
        if constexpr (FUSE_OP) {
            constexpr int TPCc = N / RPT;
            if (force_form) {
                const RoundOp<T, APPLY, 1> op{g, ck, sc, colok};
                run_passes_p2<T, LOGN, true, RB, HT, TPCc, 1, false, 0, 1, RoundOp<T, APPLY, 1>>(buf, tw, tb, col, tj, op);
            } else {
                const RoundOp<T, APPLY, 0> op{g, ck, sc, colok};
                run_passes_p2<T, LOGN, true, RB, HT, TPCc, 1, false, 0, 1, RoundOp<T, APPLY, 0>>(buf, tw, tb, col, tj, op);
            }
        } else {
        if (force_form) {
#pragma unroll PIPE ? 1 : RPT
            for (int u = 0; u < RPT; u++) {
                int n = tj + u * TPC;
                const int at = P2 ? (PIPE ? lds_at<T, RB, false>(row_base<T, RB, false>(n, col), 0) : lds_at<T, RB, false>(tb, u * TPC))
                                  : lds_index<T, RB, false>(n, col);
                cpx<T> v = buf[at];
                v.x *= sc;
                v.y *= sc;
                if (colok) v = apply_simple<T, 1>(g, n, ck, v);
                buf[at] = v;
            }
        } else {
#pragma unroll PIPE ? 1 : RPT
            for (int u = 0; u < RPT; u++) {
                int n = tj + u * TPC;
                const int at = P2 ? (PIPE ? lds_at<T, RB, false>(row_base<T, RB, false>(n, col), 0) : lds_at<T, RB, false>(tb, u * TPC))
                                  : lds_index<T, RB, false>(n, col);
                cpx<T> v = buf[at];
                v.x *= sc;
                v.y *= sc;
                if (APPLY && colok) v = apply_simple<T>(g, n, ck, v);
                buf[at] = v;
            }
        }
        __syncthreads();
        run_passes<T, LOGN, true, RB, HT>(buf, tw, col, tj);
        }
        int64_t uoff[RPT];
#pragma unroll
        for (int u = 0; u < RPT; u++) uoff[u] = uniform_offset<PIPE>((int64_t)(u * TPC) * g.out.sn);
        auto store_lines = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < RPT; u++)
                othread[uoff[u]] = buf[P2 ? lds_at<T, RB, false>(tb, u * TPC) : lds_index<T, RB, false>(tj + u * TPC, col)];
        };
        if (PIPE) {
            if (b0 + W <= g.B) {
                store_lines();
            } else {
                if (colok) store_lines();
                __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): see colfft_kernel
            }
        } else if (colok) {
            store_lines();
        }
    };
    int64_t tile = blockIdx.x;
    if (tile >= tilesB) return;
    if (PMX_COL_XCD && g.xcd) tile = xcd_tile(tile, PIPE ? (int64_t)gridDim.x : tilesB);
    __syncthreads();
    if (PIPE) {
        load_tile(tile, ld);
        do_tile(tile);
        tile += step;
    }
    for (; tile < tilesB; tile += step) do_tile(tile);
}

// ---- unit-stride real <-> half-complex row transform (the contiguous axis) ----------
// In place on rows of N = 2M reals (row pitch `pitch` complex = 2*pitch reals) <-> M+1
// complex modes.  A workgroup takes W rows, reads them as M complex z[m] = x[2m] + i x[2m+1],
// transposes them into the same LDS tile the column kernel uses (row r of the batch is
// "column" r of the tile), runs the M-point Stockham FFT and splits even/odd:
//   X[k] = (Z[k] + conj Z[M-k])/2 - (i/2) w^k (Z[k] - conj Z[M-k]),   w = exp(-2 pi i / N)
// (inverse: Z[k] = (X[k] + conj X[M-k]) + i conj(w)^k (X[k] - conj X[M-k]), then the inverse
// FFT; unnormalised like rocFFT's C2R).  One read and one write of the array.
#ifndef PMX_ROW_ONE_TILE
#define PMX_ROW_ONE_TILE 1
#endif
#ifndef PMX_ROW_LPT
#define PMX_ROW_LPT 8
#endif
// Rows whose tile and 2M twiddles fill just over half the LDS (M = 512 in double: 82 KB) keep only the first M
// twiddles — all the X <-> Z step needs; the passes take w[m + M] = -w[m] — and hold their registers to 128, so
// that two workgroups share a CU: the row passes of 1024^3 in double 3.9 -> ... ms.
#ifndef PMX_ROW_HALFTW
#define PMX_ROW_HALFTW 1
#endif
#ifndef PMX_ROW_PACK_NYQUIST
#define PMX_ROW_PACK_NYQUIST 1
#endif
#ifndef PMX_ROW_BOUND_1024
#define PMX_ROW_BOUND_1024 2
#endif
// [r4] The halo merge of the paint inside the forward row pass (pmx_rowfft_halo).  pmx_paint_binned_defer has left
// the halo cells of every tile — the part of a tile's region beyond its own T0 x T1 x T2 box — in the plan's staging
// buffer (compact numbering, Region<S>::halo_index) instead of adding them to their owners with atomics.  A canvas
// cell (x, y, z) owes up to seven of them: with a = x mod T0, b = y mod T1, c = z mod T2, every (dx, dy, dz) != 0 with
// dx = 1 only if a < S - 1 (likewise dy, dz) names the tile (tx - dx, ty - dy, tz - dz) (periodic: the one rank's whole
// mesh) and its region cell (a + dx T0, b + dy T1, c + dz T2) — unless the paint kernel carried that cell to the next
// tile of its segment in LDS instead of staging it (the walk axis: z for CIC / TSC, x for PCS; every tile but the last
// of a segment), in which case it arrived in the owner's own flush.  The row pass reads whole rows of the canvas, z
// contiguous: lane n holds cells 2n, 2n + 1 of its rows and adds what they are owed as it loads them.
struct HaloSrc {
    const void *halo;      // staging buffer of the plan (elements of the canvas type)
    int S, nt0, nt1, nt2;  // window support, tiles per axis
    int x0;                // plane of row 0 of this launch
};

// [r6] The last-axis split of a pencil transform's first transpose on the row pass itself (pmx_rowfft_split): the
// modes 0 .. M of every row cut into n ranges [e[q], e[q + 1]), range q of all rows one dense (nrows, e[q + 1] - e[q])
// array at element nrows * e[q] — the send buffer of the all-to-all over the row group (what pmx_slab_pack makes of the
// (nrows, M + 1) block in a sweep of its own).  Forward passes write that layout, inverse passes read it.
struct RowSeg {
    int n;                      // ranges (<= PMX_MAXSEG)
    int last;                   // e[n - 1]: the range the Nyquist mode M lies in starts here
    int e[PMX_MAXSEG + 1];
};

// One thread's LPT elements: column n (cells 2n, 2n + 1) of the rows yb, yb + RSTEP, ... of plane x, all inside one tile
// row (a workgroup's rows start at a multiple of their count, which divides T1); `sink(u, a0, a1)` adds to element u —
// the row's own elements, still on their way: they are only touched when a batch of staged values has arrived.  Two batches — the staged cells of this lane's own
// z tile, then (the few lanes that hold the first cells of a z tile) those of the z-faces of the tile before — and in
// each batch every load is issued into a register of its own before anything is added: a gather written per element
// (load, add, next) makes the compiler wait for memory once per row and source (0.37 -> 0.55 ms at 512^3).  The
// branch on x is uniform over the workgroup; the rows that are owed y-faces are the first S - 1 of a tile row, i.e.
// only the thread's first UY elements can be, whatever the window.
// UB: elements per batch (LPT: all at once; less where the registers of a whole batch would spill).
template <typename T, int LPT, int RSTEP, int UB, typename SINK>
__device__ __forceinline__ void halo_gather(const HaloSrc &h, int x, int yb, int n, SINK &&sink)
{
    constexpr int UY = (3 + RSTEP - 1) / RSTEP < LPT ? (3 + RSTEP - 1) / RSTEP : LPT;     // by0 + u RSTEP < S - 1 <= 3 needs u < UY
    static_assert(LPT % UB == 0, "halo_gather: batches");
    const int S1 = h.S - 1;
    const int R1 = T1 + S1, R2 = T2 + S1;
    const int NA = S1 * R1 * R2, NB = T0 * S1 * R2, HALO = NA + NB + T0 * T1 * S1;
    const bool wx = h.S >= 4 ? walk_x(4) : (h.S == 3 ? walk_x(3) : walk_x(2));
    const int tx = x / T0, ax = x % T0, ty = yb / T1, by0 = yb % T1;
    const int z = 2 * n, tz = z / T2, cz = z % T2;
    const int txm = tx == 0 ? h.nt0 - 1 : tx - 1, tym = ty == 0 ? h.nt1 - 1 : ty - 1, tzm = tz == 0 ? h.nt2 - 1 : tz - 1;
    // the tile `tw` of `ntw` along the walk axis stages (does not carry) its face
    auto last = [&](int tw, int ntw) __attribute__((always_inline)) { return tw == ntw - 1 || (tw % ZSEG) == ZSEG - 1; };
    const bool xst = ax < S1 && (!wx || last(txm, h.nt0));      // the x-face rows of this plane are staged (uniform)
    const bool fzs = cz < S1 && (wx || last(tzm, h.nt2));       // this lane's cells are owed staged z-faces
    const bool two = cz + 1 < S1;
    const T *base = (const T *)h.halo;
    const T *t10 = base + ((int64_t)txm * h.nt1 + ty) * h.nt2 * HALO, *t11 = base + ((int64_t)txm * h.nt1 + tym) * h.nt2 * HALO;
    const T *t01 = base + ((int64_t)tx * h.nt1 + tym) * h.nt2 * HALO, *t00 = base + ((int64_t)tx * h.nt1 + ty) * h.nt2 * HALO;
    // region rows (T0 + ax, by), (T0 + ax, T1 + by) of the tiles before along x (and y), (ax, T1 + by) of the tile before along y
    auto rowx = [&](int by) __attribute__((always_inline)) { return (ax * R1 + by) * R2; };
    auto rowc = [&](int by) __attribute__((always_inline)) { return (ax * R1 + by + T1) * R2; };
    auto rowy = [&](int by) __attribute__((always_inline)) { return NA + (ax * S1 + by) * R2; };
    if (!xst) {
        // Most workgroups (7 of 8 planes for CIC, 29 of 32 for PCS): no x-face.  What is left — the y-faces of the first
        // rows of a tile row, and for the lanes at the start of a z tile the z-faces — is ONE batch of loads, issued right
        // behind the row's own: the gather then costs its bytes, not a memory latency of its own.
        T y0[UY], y1[UY], w0[UY], w1[UY], z0[LPT], z1[LPT];
#pragma unroll
        for (int u = 0; u < LPT; u++) z0[u] = z1[u] = (T)0;
#pragma unroll
        for (int u = 0; u < UY; u++) y0[u] = y1[u] = w0[u] = w1[u] = (T)0;
#pragma unroll
        for (int u = 0; u < UY; u++)
            if (by0 + u * RSTEP < S1) {
                const T *p = t01 + tz * HALO + cz + rowy(by0 + u * RSTEP);
                y0[u] = p[0]; y1[u] = p[1];
            }
        if (fzs) {
#pragma unroll
            for (int u = 0; u < LPT; u++) {
                const T *p = t00 + tzm * HALO + NA + NB + (ax * T1 + by0 + u * RSTEP) * S1 + cz;
                z0[u] = p[0];
                if (two) z1[u] = p[1];
            }
#pragma unroll
            for (int u = 0; u < UY; u++)
                if (by0 + u * RSTEP < S1) {
                    const T *p = t01 + tzm * HALO + T2 + cz + rowy(by0 + u * RSTEP);
                    w0[u] = p[0];
                    if (two) w1[u] = p[1];
                }
        }
#pragma unroll
        for (int u = 0; u < LPT; u++) {
            T a0 = z0[u], a1 = z1[u];
            if (u < UY) { a0 += y0[u] + w0[u]; a1 += y1[u] + w1[u]; }
            sink(u, a0, a1);
        }
        return;
    }
    for (int pass = 0; pass < 2; pass++) {
        // pass 0: cells (cz, cz + 1) of the rows in this lane's own z tile; pass 1: (T2 + cz, T2 + cz + 1) in the z tile before
        if (pass == 1 && !fzs) break;
        const int off = pass == 0 ? tz * HALO + cz : tzm * HALO + T2 + cz;
        const bool second = pass == 0 || two;
#pragma unroll
        for (int ub = 0; ub < LPT; ub += UB) {
            if (pass == 0 && !xst && ub >= UY) break;           // (uniform) nothing but y-faces in this pass, and those are done
            T x0[UB], x1[UB], c0[UB], c1[UB], y0[UB], y1[UB], z0[UB], z1[UB];
#pragma unroll
            for (int k = 0; k < UB; k++) x0[k] = x1[k] = z0[k] = z1[k] = c0[k] = c1[k] = y0[k] = y1[k] = (T)0;
            if (xst) {
#pragma unroll
                for (int k = 0; k < UB; k++) {
                    const T *p = t10 + off + rowx(by0 + (ub + k) * RSTEP);
                    x0[k] = p[0];
                    if (second) x1[k] = p[1];
                }
#pragma unroll
                for (int k = 0; k < UB; k++)
                    if (ub + k < UY && by0 + (ub + k) * RSTEP < S1) {
                        const T *p = t11 + off + rowc(by0 + (ub + k) * RSTEP);
                        c0[k] = p[0];
                        if (second) c1[k] = p[1];
                    }
            }
#pragma unroll
            for (int k = 0; k < UB; k++)
                if (ub + k < UY && by0 + (ub + k) * RSTEP < S1) {
                    const T *p = t01 + off + rowy(by0 + (ub + k) * RSTEP);
                    y0[k] = p[0];
                    if (second) y1[k] = p[1];
                }
            if (pass == 1) {
                // (ax, by, T2 + cz) of the tile's own column: the compact z-face
#pragma unroll
                for (int k = 0; k < UB; k++) {
                    const T *p = t00 + tzm * HALO + NA + NB + (ax * T1 + by0 + (ub + k) * RSTEP) * S1 + cz;
                    z0[k] = p[0];
                    if (two) z1[k] = p[1];
                }
            }
#pragma unroll
            for (int k = 0; k < UB; k++) {
                T a0 = x0[k] + z0[k], a1 = x1[k] + z1[k];
                if (ub + k < UY) { a0 += c0[k] + y0[k]; a1 += c1[k] + y1[k]; }
                sink(ub + k, a0, a1);
            }
        }
    }
}

template <typename T, int LOGM, int RB> struct RowHalfTw {
    static constexpr int M = Len<LOGM>::N, W = RB / (int)sizeof(cpx<T>);
    // [r6] XM: the slots of the rows' Nyquist modes (inverse passes).  PACK: the power-of-two rows whose threads walk
    // along the row (2048 reals in double: a tile of 4 rows, 512 threads) keep that mode in the imaginary part of the DC
    // slot instead — both are real by definition — and need none: with the first M twiddles their tile is 80 KB to the
    // byte, two workgroups per CU
    static constexpr bool PACK = PMX_ROW_PACK_NYQUIST && LOGM < 16 && (M / PMX_ROW_LPT * W) % M != 0;
    static constexpr int XM = PACK ? 0 : W;
    static constexpr size_t full = (size_t)(M * W + 2 * M + XM) * sizeof(cpx<T>), half = (size_t)(M * W + M + XM) * sizeof(cpx<T>);
    static constexpr bool value = PMX_ROW_HALFTW && LOGM < 16 && full > 80 * 1024 && half <= 80 * 1024;
    static constexpr size_t lds = value ? half : full;
    // waves per SIMD the kernel is held to: two workgroups per CU where two tiles fit the LDS — 512 threads: 4 (128
    // registers), [r6] 1024 threads (float rows of 1024 / 2048 reals, 65-67 registers unbounded: ONE workgroup): 8 (64)
    static constexpr int NT = M / PMX_ROW_LPT * W;
    static constexpr int tight = (PMX_ROW_BOUND_1024 && NT == 1024 && lds <= 80 * 1024) ? 8 : (value ? 4 : 1);
    static constexpr int loose = value ? 4 : 1;          // (the forms that would spill under the tight bound)
};

template <typename T, int LOGM, bool INV, int RB, bool HALO = false, bool SEG = false>
__global__ void __launch_bounds__(Len<LOGM>::N / PMX_ROW_LPT * (RB / (int)sizeof(cpx<T>)),
                                  ((HALO || (SEG && INV && PMX_ROW_BOUND_1024 < 2)) ? RowHalfTw<T, LOGM, RB>::loose : RowHalfTw<T, LOGM, RB>::tight))
rowfft_kernel(cpx<T> *data_, int64_t nrows, int64_t pitch, double scale, const cpx<T> *twiddle /* length 2M */,
              int64_t rpp, int64_t plane_extra, HaloSrc hs, cpx<T> *dst_, RowSeg seg)
{
    // SEG: the complex side of the pass is the split layout of RowSeg (forward: dst_, inverse: data_), dense rows
    // without planes; the other side is laid out as always
    // dst_: where the rows are written (the same layout as data_); nullptr: in place
    // (measured and taken out again: the same gather in the 3 * 2^k / 5 * 2^k kernels — parity-green, and slower than the
    // merge kernel it replaces: 384^3 2.56 -> 2.65 ms per cycle, 768^3 20.5 -> 20.9; their row passes pay more for the
    // registers and the extra latency than the power-of-two ones)
    static_assert(!HALO || (!INV && LOGM < 16), "the halo gather rides on the forward pass of power-of-two rows");
    constexpr int M = Len<LOGM>::N;
    constexpr int W = RB / (int)sizeof(cpx<T>);   // rows per tile
    constexpr int LPT = PMX_ROW_LPT;   // row elements per thread
    constexpr int TPC = M / LPT;
    constexpr int NT = TPC * W;
    static_assert(!SEG || !HALO, "the split layout and the halo gather do not meet (pencil blocks merge their halos before)");
    extern __shared__ __align__(16) unsigned char smem[];
    cpx<T> *buf = reinterpret_cast<cpx<T> *>(smem);
    constexpr bool HT = RowHalfTw<T, LOGM, RB>::value;
    cpx<T> *tw = buf + M * W;          // 2M entries: exp(-2 pi i m / 2M) (HT: the first M)
    cpx<T> *xm = tw + (HT ? M : 2 * M);           // W entries: the Nyquist mode X[M] of every row
    const int tid0 = threadIdx.x;
    for (int n = tid0; n < (HT ? M : 2 * M); n += NT) tw[n] = twiddle[n];
    const T sc = (T)scale;
    const int64_t ntiles = (nrows + W - 1) / W;
    using Rd = Radices<LOGM>;
    // one workgroup per tile (grid = ntiles), as in the column kernel: a grid-stride loop makes the
    // optimiser carry tile-invariant values across the passes
    // (measured: 512^3 fp32 rows 3 % and 768^3 5 % faster that way; from M = 512 on the reload of the
    // 2M twiddles per tile costs more than the registers bring: 1024^3 3.5 % slower — so only below)
    constexpr bool ONE_TILE = PMX_ROW_ONE_TILE && M < 512;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += (ONE_TILE ? ntiles : (int64_t)gridDim.x)) {
        // (the lane's position is opaque per trip, as in the column kernel: what derives from it is recomputed
        // per tile instead of being carried in registers across the loop)
        int tidl = tid0;
        if (!ONE_TILE) asm volatile("" : "+v"(tidl));
        const int tid = tidl, col = tid % W, tj = tid / W;
        const int64_t r0 = tile * W;
        // rows are grouped in planes of rpp rows (a multiple of W: a tile never straddles two
        // planes) whose stride exceeds rpp*pitch by plane_extra elements
        cpx<T> *data = data_ + (rpp > 0 ? (r0 / rpp) * plane_extra : 0);
        cpx<T> *dout = dst_ ? dst_ + (rpp > 0 ? (r0 / rpp) * plane_extra : 0) : data;
        // the split side (SEG): mode k of row r, and the Nyquist mode of row r
        auto seg_at = [&](int64_t r, int k) __attribute__((always_inline)) {
            int eq = 0, eq1 = seg.e[1];
#pragma unroll
            for (int s = 1; s < PMX_MAXSEG; s++)
                if (s < seg.n && k >= seg.e[s]) { eq = seg.e[s]; eq1 = seg.e[s + 1]; }
            return nrows * eq + r * (int64_t)(eq1 - eq) + (k - eq);
        };
        auto nyquist = [&](int64_t r) __attribute__((always_inline)) {
            return nrows * seg.last + r * (int64_t)(M + 1 - seg.last) + (M - seg.last);
        };
        __syncthreads();
        if constexpr (LOGM < 16 && NT % M == 0) {
            // Power-of-two rows.  NT is a multiple of M (all but 2048 reals in double, whose tile has 4 rows), so a thread keeps ONE position along the row for all
            // its LPT elements (n = tid % M) and steps through the rows of the tile (r = tid / M + u * NT / M):
            // the swizzled LDS row and the global address are computed once per thread, the per-element part is
            // a rotation of the column slot and a workgroup-uniform row stride.
            constexpr int RSTEP = NT / M;                      // rows between successive elements of a thread
            static_assert(NT % M == 0 && RSTEP * LPT == W, "row kernel: thread -> element mapping");
            const int n = tid & (M - 1), rt = tid / M;
            const bool full = r0 + W <= nrows;                 // all but (at most) the last tile
            cpx<T> *gthread = data + ((r0 + rt) * pitch + n);
            cpx<T> *gstore = dout + ((r0 + rt) * pitch + n);
            // strides between a thread's successive rows, elements: uniform but on the split side, where the range the
            // thread's position lies in decides
            int64_t ldstep = (int64_t)RSTEP * pitch, ststep = (int64_t)RSTEP * pitch;
            if constexpr (SEG) {
                int eq = 0, eq1 = seg.e[1];
#pragma unroll
                for (int s = 1; s < PMX_MAXSEG; s++)
                    if (s < seg.n && n >= seg.e[s]) { eq = seg.e[s]; eq1 = seg.e[s + 1]; }
                const int wq = eq1 - eq;
                const int64_t at0 = nrows * eq + (r0 + rt) * wq + (n - eq);
                if (INV) { gthread = data_ + at0; ldstep = (int64_t)RSTEP * wq; }
                else { gstore = dst_ + at0; ststep = (int64_t)RSTEP * wq; }
            }
            const RowBase<T, RB> nb_ = row_base<T, RB>(n, 0);
            auto at = [&](const RowBase<T, RB> &b, int r) __attribute__((always_inline)) {
                return b.e[0] + ((b.cr + r) & (W - 1));
            };
            cpx<T> ld[LPT];
#pragma unroll
            for (int u = 0; u < LPT; u++) ld[u] = cpx<T>{0, 0};
            if (full) {
#pragma unroll
                for (int u = 0; u < LPT; u++) ld[u] = gthread[u * ldstep];
            } else {
#pragma unroll
                for (int u = 0; u < LPT; u++)
                    if (r0 + rt + u * RSTEP < nrows) ld[u] = gthread[u * ldstep];
            }
            if (INV && tid < W) {
                if constexpr (SEG) xm[tid] = (r0 + tid < nrows) ? data_[nyquist(r0 + tid)] : cpx<T>{0, 0};
                else xm[tid] = (r0 + tid < nrows) ? data[(r0 + tid) * pitch + M] : cpx<T>{0, 0};
            }
            if constexpr (HALO) {
                // rows r0 .. r0 + W - 1 lie in one plane and one tile row (rpp is a multiple of W, W divides T1; whole
                // planes: every tile is full): x is the workgroup's, y the row's
                static_assert(T1 % W == 0, "row kernel: a tile of rows inside one tile row of the mesh");
                if constexpr (!HT)
                    halo_gather<T, LPT, RSTEP, LPT>(hs, hs.x0 + (int)(r0 / rpp), (int)(r0 % rpp) + rt, n,
                                                    [&](int u, T a0, T a1) __attribute__((always_inline)) { ld[u].x += a0; ld[u].y += a1; });
            }
#pragma unroll
            for (int u = 0; u < LPT; u++) buf[at(nb_, rt + u * RSTEP)] = ld[u];
            if constexpr (HALO && HT) {
                // the kernels that live inside 128 registers (two 512-thread workgroups per CU): the row's own elements
                // go to LDS first, the staged values are gathered half a batch at a time and added there — every
                // thread to the slots it has just written, before the barrier
                halo_gather<T, LPT, RSTEP, LPT / 2>(hs, hs.x0 + (int)(r0 / rpp), (int)(r0 % rpp) + rt, n,
                                                    [&](int u, T a0, T a1) __attribute__((always_inline)) {
                                                        cpx<T> v = buf[at(nb_, rt + u * RSTEP)];
                                                        v.x += a0; v.y += a1;
                                                        buf[at(nb_, rt + u * RSTEP)] = v;
                                                    });
            }
            __syncthreads();
            if (INV) {
                // X -> Z, pairs (k, M-k) handled together, in place: k in [1, M/2) by all threads (k fixed per
                // thread), k = 0 (paired with the Nyquist mode kept in xm) and k = M/2 (its own partner) likewise
                constexpr int H = M / 2;
                constexpr int PSTEP = NT / H;
                const int k = tid & (H - 1), rp = tid / H;
                const int k2 = M - k;
                const RowBase<T, RB> bk = row_base<T, RB>(k, 0), bq = row_base<T, RB>(k == 0 ? H : k2, 0);
                cpx<T> w = tw[k];
                w.y = -w.y;                                  // conj(w)^k = exp(+2 pi i k / N)
#pragma unroll
                for (int i = 0; i < W / PSTEP; i++) {
                    const int r = rp + i * PSTEP;
                    cpx<T> xk = buf[at(bk, r)];
                    cpx<T> xq = (k == 0) ? xm[r] : buf[at(bq, r)];
                    if (k == 0) { xk.y = 0; xq.y = 0; }
                    cpx<T> A = {xk.x + xq.x, xk.y - xq.y}, D = {xk.x - xq.x, xk.y + xq.y};
                    cpx<T> B = cmul(w, D);
                    buf[at(bk, r)] = {A.x - B.y, A.y + B.x};            // A + i B
                    if (k != 0) {
                        cpx<T> Ac = {A.x, -A.y};
                        cpx<T> Dc = {-D.x, D.y};
                        cpx<T> wc = {-w.x, w.y};
                        cpx<T> Bc = cmul(wc, Dc);
                        buf[at(bq, r)] = {Ac.x - Bc.y, Ac.y + Bc.x};
                    } else {
                        // k = 0's thread also takes k = M/2, whose partner is itself (only Z[M/2] is written)
                        cpx<T> xh = buf[at(bq, r)];
                        cpx<T> Ah = {xh.x + xh.x, xh.y - xh.y}, Dh = {xh.x - xh.x, xh.y + xh.y};
                        cpx<T> wh = tw[H];
                        wh.y = -wh.y;
                        cpx<T> Bh = cmul(wh, Dh);
                        buf[at(bq, r)] = {Ah.x - Bh.y, Ah.y + Bh.x};
                    }
                }
                __syncthreads();
            }
            {
                const RowBase<T, RB> tb = row_base<T, RB>(tj, col);
                run_passes_p2<T, LOGM, INV, RB, HT, TPC, 2, true, 0, 1>(buf, tw, tb, col, tj);
            }
            if (!INV) {
                // Z -> X for k = 0..M-1 (k = n: fixed per thread), then the Nyquist mode and the rest of its line
                const RowBase<T, RB> bq = row_base<T, RB>(n == 0 ? 0 : M - n, 0);
                const cpx<T> w = tw[n];
                auto mode = [&](cpx<T> zk, cpx<T> zq, cpx<T> wk) __attribute__((always_inline)) {
                    cpx<T> E = {(T)0.5 * (zk.x + zq.x), (T)0.5 * (zk.y - zq.y)};       // (zk + conj zq)/2
                    cpx<T> D = {(T)0.5 * (zk.x - zq.x), (T)0.5 * (zk.y + zq.y)};       // (zk - conj zq)/2
                    cpx<T> wd = cmul(wk, D);
                    cpx<T> X = {E.x + wd.y, E.y - wd.x};                               // X = E - i * w * D
                    X.x *= sc; X.y *= sc;
                    return X;
                };
#pragma unroll
                for (int u = 0; u < LPT; u++) {
                    const int r = rt + u * RSTEP;
                    cpx<T> X = mode(buf[at(nb_, r)], buf[at(bq, r)], w);
                    if (full || r0 + r < nrows) gstore[u * ststep] = X;
                }
                constexpr int LINE = 128 / (int)sizeof(cpx<T>);
                const int tail = (!SEG && pitch >= M + LINE) ? LINE : 1;       // elements M .. M + tail - 1 of every row
                const RowBase<T, RB> b0 = row_base<T, RB>(0, 0);
                for (int q = tid; q < W * tail; q += NT) {
                    const int r = q / tail, j = q - r * tail;
                    if (r0 + r < nrows) {
                        cpx<T> z0 = buf[at(b0, r)];
                        if constexpr (SEG) dst_[nyquist(r0 + r)] = mode(z0, z0, cpx<T>{(T)-1, (T)0});
                        else dout[(r0 + r) * pitch + M + j] = (j == 0) ? mode(z0, z0, cpx<T>{(T)-1, (T)0}) : cpx<T>{(T)0, (T)0};
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < LPT; u++) {
                    const int r = rt + u * RSTEP;
                    cpx<T> v = buf[at(nb_, r)];
                    v.x *= sc; v.y *= sc;
                    if (full || r0 + r < nrows) gstore[u * ststep] = v;
                }
            }
            continue;
        }
        // load W rows, consecutive lanes along the row
        cpx<T> ld[LPT];
#pragma unroll
        for (int u = 0; u < LPT; u++) {
            int flat = tid + u * NT;
            int r = flat / M, n = flat % M;
            if constexpr (SEG && INV) ld[u] = (r0 + r < nrows) ? data_[seg_at(r0 + r, n)] : cpx<T>{0, 0};
            else ld[u] = (r0 + r < nrows) ? data[(r0 + r) * pitch + n] : cpx<T>{0, 0};
        }
        constexpr bool PACK = RowHalfTw<T, LOGM, RB>::PACK;
        if constexpr (PACK) {
            // the DC and the Nyquist mode of a real row are real (see below): the Nyquist mode travels in the DC slot's
            // imaginary part, no slots of its own.  flat = tid + u NT is a multiple of M for thread 0 alone (NT divides M)
            static_assert(M % NT == 0 && (M / NT) * (W - 1) < LPT, "row kernel: thread 0 holds the DC slots of the tile's rows");
            if (INV && tid == 0) {
#pragma unroll
                for (int r = 0; r < W; r++)
                    if (r0 + r < nrows) {
                        if constexpr (SEG) ld[r * (M / NT)].y = data_[nyquist(r0 + r)].x;
                        else ld[r * (M / NT)].y = data[(r0 + r) * pitch + M].x;
                    }
            }
        } else if (INV && tid < W) {
            if constexpr (SEG) xm[tid] = (r0 + tid < nrows) ? data_[nyquist(r0 + tid)] : cpx<T>{0, 0};
            else xm[tid] = (r0 + tid < nrows) ? data[(r0 + tid) * pitch + M] : cpx<T>{0, 0};
        }
#pragma unroll
        for (int u = 0; u < LPT; u++) {
            int flat = tid + u * NT;
            buf[lds_index<T, RB>(flat % M, flat / M)] = ld[u];
        }
        __syncthreads();
        if (INV) {
            // X -> Z, pairs (k, M-k) handled together, in place
            for (int q = tid; q < W * (M / 2 + 1); q += NT) {
                int r = q / (M / 2 + 1), k = q % (M / 2 + 1);
                int k2 = M - k;
                cpx<T> xk = buf[lds_index<T, RB>(k, r)];
                cpx<T> xq;
                if constexpr (PACK) xq = (k == 0) ? cpx<T>{xk.y, 0} : buf[lds_index<T, RB>(k2, r)];
                else xq = (k == 0) ? xm[r] : buf[lds_index<T, RB>(k2, r)];
                // the DC and Nyquist modes of a real row are real: their imaginary parts are
                // ignored, as FFTW's c2r (behind PFFT) and numpy.fft.irfft do — it matters for
                // spectra that are not exactly Hermitian, e.g. after i k / k^2 on the Nyquist planes
                if (k == 0) { xk.y = 0; xq.y = 0; }
                // A = xk + conj(xq), D = xk - conj(xq)
                cpx<T> A = {xk.x + xq.x, xk.y - xq.y}, D = {xk.x - xq.x, xk.y + xq.y};
                cpx<T> w = tw[k];
                w.y = -w.y;                                  // conj(w)^k = exp(+2 pi i k / N)
                cpx<T> B = cmul(w, D);
                buf[lds_index<T, RB>(k, r)] = {A.x - B.y, A.y + B.x};            // A + i B
                if (k != 0 && k2 != k) {
                    // Z[M-k] = conj(A) + i conj(w')... computed from the same pair:
                    // A' = xq + conj(xk) = conj(A),  D' = xq - conj(xk) = -conj(D),  w' = exp(+2 pi i (M-k)/N) = -conj(w)
                    cpx<T> Ac = {A.x, -A.y};
                    cpx<T> Dc = {-D.x, D.y};
                    cpx<T> wc = {-w.x, w.y};
                    cpx<T> Bc = cmul(wc, Dc);
                    buf[lds_index<T, RB>(k2, r)] = {Ac.x - Bc.y, Ac.y + Bc.x};
                }
            }
            __syncthreads();
        }
        int Ns = 1;
        if constexpr (LOGM < 16) {
            const RowBase<T, RB> tb = row_base<T, RB>(tj, col);
            run_passes_p2<T, LOGM, INV, RB, HT, TPC, 2, true, 0, 1>(buf, tw, tb, col, tj);
        } else {
        if (Rd::r[0] == 8) stockham_pass<T, INV, 8, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
        Ns *= Rd::r[0];
        if (Rd::n > 1) {
            if (Rd::r[1] == 8) stockham_pass<T, INV, 8, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            else if (Rd::r[1] == 4) stockham_pass<T, INV, 4, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            else if (Rd::r[1] == 3) stockham_pass<T, INV, 3, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            else if (Rd::r[1] == 5) stockham_pass<T, INV, 5, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            Ns *= Rd::r[1];
        }
        if (Rd::n > 2) {
            if (Rd::r[2] == 8) stockham_pass<T, INV, 8, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            else if (Rd::r[2] == 4) stockham_pass<T, INV, 4, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            else if (Rd::r[2] == 3) stockham_pass<T, INV, 3, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            else if (Rd::r[2] == 5) stockham_pass<T, INV, 5, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            Ns *= Rd::r[2];
        }
        if (Rd::n > 3) {
            if (Rd::r[3] == 8) stockham_pass<T, INV, 8, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            else if (Rd::r[3] == 4) stockham_pass<T, INV, 4, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            else if (Rd::r[3] == 3) stockham_pass<T, INV, 3, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            else if (Rd::r[3] == 5) stockham_pass<T, INV, 5, RB>(buf, tw, M, Ns, TPC, col, tj, 2);
            Ns *= Rd::r[3];
        }
        }
        if (!INV) {
            // Z -> X for k = 0..M, consecutive lanes along the row.  The Nyquist mode k = M starts a new
            // 128-byte line of the row; where the row is padded (one rank: pitch = M + 8 in double) the
            // rest of that line is written too (zeros): a whole line instead of a 16-byte piece that the
            // memory side would have to merge into the old one.
            auto mode = [&](int r, int k) __attribute__((always_inline)) {
                cpx<T> zk = buf[lds_index<T, RB>(k == M ? 0 : k, r)];
                cpx<T> zq = buf[lds_index<T, RB>(k == 0 ? 0 : M - k, r)];
                cpx<T> E = {(T)0.5 * (zk.x + zq.x), (T)0.5 * (zk.y - zq.y)};       // (zk + conj zq)/2
                cpx<T> D = {(T)0.5 * (zk.x - zq.x), (T)0.5 * (zk.y + zq.y)};       // (zk - conj zq)/2
                cpx<T> w = (k == M) ? cpx<T>{(T)-1, (T)0} : tw[k];
                cpx<T> wd = cmul(w, D);
                // X = E - i * w * D
                cpx<T> X = {E.x + wd.y, E.y - wd.x};
                X.x *= sc; X.y *= sc;
                return X;
            };
            for (int q = tid; q < W * M; q += NT) {
                const int r = q / M, k = q % M;
                if constexpr (SEG) { if (r0 + r < nrows) dst_[seg_at(r0 + r, k)] = mode(r, k); }
                else if (r0 + r < nrows) dout[(r0 + r) * pitch + k] = mode(r, k);
            }
            constexpr int LINE = 128 / (int)sizeof(cpx<T>);
            const int tail = (!SEG && pitch >= M + LINE) ? LINE : 1;       // elements M .. M + tail - 1 of every row
            for (int q = tid; q < W * tail; q += NT) {
                const int r = q / tail, j = q - r * tail;
                if constexpr (SEG) { if (r0 + r < nrows) dst_[nyquist(r0 + r)] = mode(r, M); }
                else if (r0 + r < nrows) dout[(r0 + r) * pitch + M + j] = (j == 0) ? mode(r, M) : cpx<T>{(T)0, (T)0};
            }
        } else {
#pragma unroll
            for (int u = 0; u < LPT; u++) {
                int flat = tid + u * NT;
                int r = flat / M, n = flat % M;
                if (r0 + r < nrows) {
                    cpx<T> v = buf[lds_index<T, RB>(n, r)];
                    v.x *= sc; v.y *= sc;
                    dout[(r0 + r) * pitch + n] = v;
                }
            }
        }
    }
}

// pmx_colfft_configure: persistent (prefetching) column passes, or one workgroup per tile
#if PMX_COLFFT_PART == 2
extern std::atomic<int> g_persistent;
#elif PMX_COLFFT_PART == 1
std::atomic<int> g_persistent{1};
#else
static std::atomic<int> g_persistent{1};
#endif

static int compute_units()
{
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// twiddle tables exp(-2 pi i m / N), one per (N, precision, device), created on first use
struct TwKey { int n, es, dev; bool operator<(const TwKey &o) const { return n != o.n ? n < o.n : (es != o.es ? es < o.es : dev < o.dev); } };
static std::map<TwKey, void *> g_tw;
static std::mutex g_tw_mutex;

static int get_twiddles(int N, int es, void **out, hipStream_t st)
{
    int dev = 0;
    PMX_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    TwKey key{N, es, dev};
    auto it = g_tw.find(key);
    if (it != g_tw.end()) { *out = it->second; return PMX_OK; }
    void *d = nullptr;
    PMX_HIP_CHECK(hipMalloc(&d, (size_t)N * 2 * es));
    // The second half of the table is the negated first half BY CONSTRUCTION (w[m + N/2] = -w[m]; libm's values of
    // the two differ in the last bit here and there): the kernels that keep only half the table in LDS (HalfTw,
    // RowHalfTw) then compute exactly what the others do — the round-trip kernel of N = 1024 in double, on its
    // narrow tiles, stays bit-identical to the two full-table passes it replaces.  (All lengths are even.)
    std::vector<double> w(2 * N);
    for (int m = 0; m < N; m++) {
        const int q = m < N / 2 ? m : m - N / 2;
        const double sg = m < N / 2 ? 1.0 : -1.0;
        w[2 * m] = sg * cos(-2.0 * M_PI * q / N);
        w[2 * m + 1] = sg * sin(-2.0 * M_PI * q / N);
    }
    if (es == 8) {
        PMX_HIP_CHECK(hipMemcpy(d, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    } else {
        std::vector<float> h(2 * N);
        for (int m = 0; m < 2 * N; m++) h[m] = (float)w[m];
        PMX_HIP_CHECK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    g_tw[key] = d;
    *out = d;
    return PMX_OK;
}

// tables of the finite-difference gradient factor D over the global index of one axis (ColGeom::dtab), one per
// (mesh side, box side, device), created on first use
struct DtKey { int64_t n; double box; int dev; bool operator<(const DtKey &o) const { return n != o.n ? n < o.n : (box != o.box ? box < o.box : dev < o.dev); } };
static std::map<DtKey, double *> g_dtab;

static int gradient_table(ColGeom &g, const pmx_transfer *t, const int64_t *nmesh, const double *boxsize)
{
    g.dtab = nullptr;
    if (!t || t->grad_dir < 0 || t->grad_kind == 0) return PMX_OK;
    const int d = t->grad_dir;
    int dev = 0;
    PMX_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_tw_mutex);
    DtKey key{nmesh[d], boxsize[d], dev};
    auto it = g_dtab.find(key);
    if (it != g_dtab.end()) { g.dtab = it->second; return PMX_OK; }
    const int64_t N = nmesh[d];
    std::vector<double> h((size_t)N);
    const double C = boxsize[d] / (double)N;
    for (int64_t gi = 0; gi < N; gi++) {
        // the wavenumber exactly as kcoord() forms it, then D as pmx_transfer.hip does (nbody.py:166-169)
        double wi = (double)gi;
        if (gi >= N / 2) wi -= (double)N;
        wi *= 2 * M_PI / (double)N;
        const double k = wi * ((double)N / boxsize[d]);
        const double w = k * C;
        h[(size_t)gi] = (1.0 / C) * (1.0 / 6.0) * (8.0 * sin(w) - sin(2.0 * w));
    }
    double *dptr = nullptr;
    PMX_HIP_CHECK(hipMalloc((void **)&dptr, (size_t)N * sizeof(double)));
    PMX_HIP_CHECK(hipMemcpy(dptr, h.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    g_dtab[key] = dptr;
    g.dtab = dptr;
    return PMX_OK;
}

template <typename T, int LOGN, int RB, bool RM>
static int launch_colfft_rm(const ColGeom &g, const void *src, void *dst, const void *tw, bool inverse, bool apply,
                            hipStream_t st)
{
    constexpr int N = Len<LOGN>::N;
    constexpr int W = RB / (int)sizeof(cpx<T>);
    constexpr int NT = N / Rpt<T, LOGN>::value * W;
    static_assert(NT <= 1024 && NT % 64 == 0, "column kernel: workgroup size");
    size_t lds = (size_t)(N * W + (HalfTw<T, LOGN, RB>::value ? N / 2 : N)) * sizeof(cpx<T>);
    int64_t tiles = g.A * ((g.B + W - 1) / W);
    PMX_REQUIRE(tiles < (1ll << 31), PMX_EUNSUPPORTED, "more than 2^31 tiles in one column pass");
    unsigned grid = (unsigned)((N < PMX_COL_STRIDE_FROM) ? tiles : (tiles < 256 * 16 ? tiles : 256 * 16));
    if (ColPipe<T, LOGN, RB>::value && !(apply && sizeof(T) == 4) && g_persistent.load(std::memory_order_relaxed)) {
        // persistent workgroups, one per CU (the tile leaves no room for a second); with grid = tiles the same
        // kernel runs one tile per workgroup (its loop makes one trip, nothing is prefetched)
        const int64_t cus = compute_units();
        grid = (unsigned)(tiles < cus ? tiles : cus);
    } else if (ColPipe<T, LOGN, RB>::value && !(apply && sizeof(T) == 4)) {
        grid = (unsigned)tiles;
    }
    // tiles in XCD order where the lines of either side do not start on 128-byte boundaries (see xcd_tile)
    ColGeom gx = g;
    {
        constexpr int64_t LINE = 128 / (int64_t)sizeof(cpx<T>);
        auto ragged = [&](const ColAddr &a) { return a.sn % LINE != 0 || (g.A > 1 && a.sa % LINE != 0) || (a.cw > 0 && (a.cw % LINE != 0 || a.cpitch % LINE != 0)); };
        // (and where a tile's row segments are half a line — RB = 64, the 2048-point passes: the other half is the next tile's)
        gx.xcd = PMX_COL_XCD >= 2 || (PMX_COL_XCD == 1 && (RB < 128 || ragged(g.in) || ragged(g.out)));
    }
#define LAUNCH(INV, AP)                                                                                        \
    do {                                                                                                       \
        auto k = colfft_kernel<T, LOGN, INV, AP, RB, RM>;                                                      \
        PMX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k<<<grid, NT, lds, st>>>(gx, (const cpx<T> *)src, (cpx<T> *)dst, (const cpx<T> *)tw);                  \
    } while (0)
    if (inverse) { if (apply) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (apply) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

template <typename T, int LOGN, int RB>
static int launch_colfft(const ColGeom &g, const void *src, void *dst, const void *tw, bool inverse, bool apply,
                         hipStream_t st)
{
    if (g.in.cw > 0 || g.out.cw > 0) return launch_colfft_rm<T, LOGN, RB, true>(g, src, dst, tw, inverse, apply, st);
    return launch_colfft_rm<T, LOGN, RB, false>(g, src, dst, tw, inverse, apply, st);
}

// (A 256-byte-row variant, RB = 256 with one workgroup per CU, was measured for the axis-0
// pass, whose rows are a whole plane apart: no gain at 512^3 — 1.42 vs 1.44 ms forward, and
// the fused-transfer pass got slower — so only the 128-byte tiles are instantiated.)
#ifndef PMX_RB_D1024
#define PMX_RB_D1024 128
#endif
#ifndef PMX_RB768
#define PMX_RB768 (sizeof(T) == 4 ? 64 : 128)
#endif
template <typename T>
PMX_DISPATCH int dispatch_logn(const ColGeom &g, const void *src, void *dst, const void *tw, bool inverse, bool apply,
                         hipStream_t st)
{
    switch (g.logN) {
    case 6: return launch_colfft<T, 6, 128>(g, src, dst, tw, inverse, apply, st);
    case 7: return launch_colfft<T, 7, 128>(g, src, dst, tw, inverse, apply, st);
    case 8: return launch_colfft<T, 8, 128>(g, src, dst, tw, inverse, apply, st);
    case 9: return launch_colfft<T, 9, 128>(g, src, dst, tw, inverse, apply, st);
    case 10: return launch_colfft<T, 10, (sizeof(T) == 8 ? PMX_RB_D1024 : 128)>(g, src, dst, tw, inverse, apply, st);
    case 11: return launch_colfft<T, 11, 64>(g, src, dst, tw, inverse, apply, st);
    // 3 * 2^k: 8 lines per thread in both precisions; the row width keeps the workgroup within
    // 1024 threads and the tile within the LDS
    case 22: return launch_colfft<T, 22, 128>(g, src, dst, tw, inverse, apply, st);
    case 23: return launch_colfft<T, 23, 128>(g, src, dst, tw, inverse, apply, st);
    case 24: return launch_colfft<T, 24, PMX_RB768>(g, src, dst, tw, inverse, apply, st);
    case 25:
        if constexpr (sizeof(T) == 8) return launch_colfft<T, 25, 64>(g, src, dst, tw, inverse, apply, st);
        break;
    // 5 * 2^k
    case 38: return launch_colfft<T, 38, 128>(g, src, dst, tw, inverse, apply, st);
    case 39: return launch_colfft<T, 39, (sizeof(T) == 4 ? 64 : 128)>(g, src, dst, tw, inverse, apply, st);
    case 40:
        if constexpr (sizeof(T) == 8) return launch_colfft<T, 40, 64>(g, src, dst, tw, inverse, apply, st);
        break;
    }
    set_error("pmx_colfft: length code %d is not built", g.logN);
    return PMX_EUNSUPPORTED;
}
#if PMX_COLFFT_PART == 1
extern template int dispatch_logn<float>(const ColGeom &, const void *, void *, const void *, bool, bool, hipStream_t);
#elif PMX_COLFFT_PART == 2
template int dispatch_logn<float>(const ColGeom &, const void *, void *, const void *, bool, bool, hipStream_t);
#endif

template <typename T, int LOGM, int RB = 128>
static int launch_rowfft(void *data, int64_t nrows, int64_t pitch, double scale, const void *tw, bool inverse,
                         int64_t rpp, int64_t plane_extra, hipStream_t st, const HaloSrc *halo = nullptr, void *dst = nullptr,
                         const RowSeg *seg = nullptr)
{
    constexpr int M = Len<LOGM>::N;
    constexpr int W = RB / (int)sizeof(cpx<T>);
    constexpr int NT = M / PMX_ROW_LPT * W;
    static_assert(NT <= 1024 && NT % 64 == 0, "row kernel: workgroup size");
    size_t lds = RowHalfTw<T, LOGM, RB>::lds;
    int64_t tiles = (nrows + W - 1) / W;
    PMX_REQUIRE(tiles < (1ll << 31), PMX_EUNSUPPORTED, "more than 2^31 row tiles in one pass");
    unsigned grid = (unsigned)((PMX_ROW_ONE_TILE && M < 512) ? tiles : (tiles < 256 * 64 ? tiles : 256 * 64));
    const HaloSrc none = {nullptr, 0, 0, 0, 0, 0};
    RowSeg noseg;
    noseg.n = 0;
    if (seg) {
        {
            PMX_REQUIRE(!halo && dst != nullptr && dst != data, PMX_EINVAL, "the split row pass works out of place");
            if (inverse) {
                auto k = rowfft_kernel<T, LOGM, true, RB, false, true>;
                PMX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                k<<<grid, NT, lds, st>>>((cpx<T> *)data, nrows, pitch, scale, (const cpx<T> *)tw, rpp, plane_extra, none, (cpx<T> *)dst, *seg);
            } else {
                auto k = rowfft_kernel<T, LOGM, false, RB, false, true>;
                PMX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                k<<<grid, NT, lds, st>>>((cpx<T> *)data, nrows, pitch, scale, (const cpx<T> *)tw, rpp, plane_extra, none, (cpx<T> *)dst, *seg);
            }
        }
    } else if (halo) {
        if constexpr (LOGM < 16 && NT % M == 0) {
            PMX_REQUIRE(!inverse && rpp > 0, PMX_EINVAL, "the halo gather rides on the forward pass over whole planes");
            auto k = rowfft_kernel<T, LOGM, false, RB, true>;
            PMX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            k<<<grid, NT, lds, st>>>((cpx<T> *)data, nrows, pitch, scale, (const cpx<T> *)tw, rpp, plane_extra, *halo, (cpx<T> *)dst, noseg);
        } else {
            set_error("pmx_rowfft_halo: row length not built");
            return PMX_EUNSUPPORTED;
        }
    } else if (inverse) {
        auto k = rowfft_kernel<T, LOGM, true, RB>;
        PMX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k<<<grid, NT, lds, st>>>((cpx<T> *)data, nrows, pitch, scale, (const cpx<T> *)tw, rpp, plane_extra, none, (cpx<T> *)dst, noseg);
    } else {
        auto k = rowfft_kernel<T, LOGM, false, RB>;
        PMX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k<<<grid, NT, lds, st>>>((cpx<T> *)data, nrows, pitch, scale, (const cpx<T> *)tw, rpp, plane_extra, none, (cpx<T> *)dst, noseg);
    }
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

template <typename T>
PMX_DISPATCH int dispatch_logm(int logm, void *data, int64_t nrows, int64_t pitch, double scale, const void *tw,
                         bool inverse, int64_t rpp, int64_t plane_extra, hipStream_t st, const HaloSrc *halo, void *dst,
                         const RowSeg *seg)
{
    switch (logm) {
    case 6: return launch_rowfft<T, 6>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, halo, dst, seg);
    case 7: return launch_rowfft<T, 7>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, halo, dst, seg);
    case 8: return launch_rowfft<T, 8>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, halo, dst, seg);
    case 9: return launch_rowfft<T, 9>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, halo, dst, seg);
    // 2048 reals: 64-byte tile rows (8 float / 4 double rows per workgroup) keep 1024 / 512 threads
    case 10: return launch_rowfft<T, 10, 64>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, halo, dst, seg);
    // n = 384, 768, 1536 reals: M = 192, 384, 768
    case 22: return launch_rowfft<T, 22>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, nullptr, dst, seg);
    case 23: return launch_rowfft<T, 23>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, nullptr, dst, seg);
    case 24: return launch_rowfft<T, 24, 64>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, nullptr, dst, seg);
    // n = 640, 1280 reals: M = 320, 640
    case 38: return launch_rowfft<T, 38>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, nullptr, dst, seg);
    case 39: return launch_rowfft<T, 39, 64>(data, nrows, pitch, scale, tw, inverse, rpp, plane_extra, st, nullptr, dst, seg);
    }
    set_error("pmx_rowfft: length code %d is not built", logm);
    return PMX_EUNSUPPORTED;
}
#if PMX_COLFFT_PART == 1
extern template int dispatch_logm<float>(int, void *, int64_t, int64_t, double, const void *, bool, int64_t, int64_t, hipStream_t, const HaloSrc *, void *, const RowSeg *);
#elif PMX_COLFFT_PART == 2
template int dispatch_logm<float>(int, void *, int64_t, int64_t, double, const void *, bool, int64_t, int64_t, hipStream_t, const HaloSrc *, void *, const RowSeg *);
#endif

}  // namespace pmx

using namespace pmx;

// PMX_OK if the real row transform of length n (n reals <-> n/2+1 modes) is built
#if PMX_COLFFT_PART != 2
extern "C" int pmx_rowfft_supported(int64_t n, int32_t elsize)
{
    if (elsize != 4 && elsize != 8) return PMX_EINVAL;
    int lc = length_code(n);
    if (lc < 0) return PMX_EUNSUPPORTED;
    if (lc < 16) return (n >= 128 && n <= 2048) ? PMX_OK : PMX_EUNSUPPORTED;
    if (lc < 32) return (n >= 384 && n <= 1536) ? PMX_OK : PMX_EUNSUPPORTED;      // 3 * 2^k: 384, 768, 1536
    return (n == 640 || n == 1280) ? PMX_OK : PMX_EUNSUPPORTED;                   // 5 * 2^k
}
#endif

// In-place real <-> half-complex transform of `nrows` rows of n reals (inverse = 0: r2c,
// n reals -> n/2+1 modes; 1: c2r), row pitch `pitch` COMPLEX elements (>= n/2+1), result
// multiplied by `scale`; unnormalised in both directions.
#if PMX_COLFFT_PART != 2
static int rowfft_any(int32_t elsize, int32_t inverse, void *data, void *dst, int64_t nrows, int64_t n, int64_t pitch,
                      double scale, int64_t rows_per_plane, int64_t plane_pitch, void *stream)
{
    int rc = pmx_rowfft_supported(n, elsize);
    if (rc) { set_error("pmx_rowfft: unsupported length %lld", (long long)n); return rc; }
    PMX_REQUIRE(data != nullptr && nrows >= 0 && pitch >= n / 2 + 1, PMX_EINVAL, "bad arguments");
    int64_t plane_extra = 0;
    if (rows_per_plane > 0) {
        // row r lives at (r / rows_per_plane) * plane_pitch + (r % rows_per_plane) * pitch
        PMX_REQUIRE(plane_pitch >= rows_per_plane * pitch, PMX_EINVAL, "plane_pitch smaller than a dense plane");
        PMX_REQUIRE(rows_per_plane % (128 / (2 * elsize)) == 0, PMX_EUNSUPPORTED,
                    "rows_per_plane must be a multiple of the rows of a tile");
        plane_extra = plane_pitch - rows_per_plane * pitch;
    }
    if (nrows == 0) return PMX_OK;
    hipStream_t st = (hipStream_t)stream;
    void *tw = nullptr;
    rc = get_twiddles((int)n, elsize, &tw, st);
    if (rc) return rc;
    int logm = length_code(n / 2);
    if (dst == data) dst = nullptr;
    if (elsize == 8)
        return dispatch_logm<double>(logm, data, nrows, pitch, scale, tw, inverse != 0, rows_per_plane, plane_extra, st, nullptr, dst, nullptr);
    return dispatch_logm<float>(logm, data, nrows, pitch, scale, tw, inverse != 0, rows_per_plane, plane_extra, st, nullptr, dst, nullptr);
}

extern "C" int pmx_rowfft(int32_t elsize, int32_t inverse, void *data, int64_t nrows, int64_t n, int64_t pitch,
                          double scale, int64_t rows_per_plane, int64_t plane_pitch, void *stream)
{
    return rowfft_any(elsize, inverse, data, nullptr, nrows, n, pitch, scale, rows_per_plane, plane_pitch, stream);
}

// the same pass from `src` into `dst` (same layout; see pmx_colfft_to)
extern "C" int pmx_rowfft_to(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t nrows, int64_t n,
                             int64_t pitch, double scale, int64_t rows_per_plane, int64_t plane_pitch, void *stream)
{
    PMX_REQUIRE(dst != nullptr, PMX_EINVAL, "dst is NULL");
    return rowfft_any(elsize, inverse, (void *)src, dst, nrows, n, pitch, scale, rows_per_plane, plane_pitch, stream);
}

// PMX_OK if the forward row pass of n reals can gather the staged halos of a paint (pmx_rowfft_halo): the
// power-of-two rows whose threads keep one position along the row (all but 2048 reals in double)
extern "C" int pmx_rowfft_halo_supported(int64_t n, int32_t elsize)
{
    int rc = pmx_rowfft_supported(n, elsize);
    if (rc) return rc;
    // power-of-two rows (the other lengths: measured, a loss), and not 2048 reals in double, whose 64-byte tiles hold
    // 4 rows: threads that walk along the row, another element mapping
    if (length_code(n) >= 16) return PMX_EUNSUPPORTED;
    if (n == 2048 && elsize == 8) return PMX_EUNSUPPORTED;
    return PMX_OK;
}

extern "C" int pmx_rowfft_halo(int32_t elsize, void *data, void *dst, int64_t nrows, int64_t n, int64_t pitch, double scale,
                               int64_t rows_per_plane, int64_t plane_pitch, pmx_binplan *plan, const void *canvas,
                               int64_t x0, int32_t last, void *stream)
{
    if (dst == data) dst = nullptr;
    int rc = pmx_rowfft_halo_supported(n, elsize);
    if (rc) { set_error("pmx_rowfft_halo: unsupported length %lld", (long long)n); return rc; }
    PMX_REQUIRE(data != nullptr && nrows >= 0 && pitch >= n / 2 + 1, PMX_EINVAL, "bad arguments");
    PMX_REQUIRE(rows_per_plane > 0 && plane_pitch >= rows_per_plane * pitch, PMX_EINVAL, "planes of rows_per_plane rows");
    PMX_REQUIRE(rows_per_plane % (128 / (2 * elsize)) == 0, PMX_EUNSUPPORTED,
                "rows_per_plane must be a multiple of the rows of a tile");
    HaloSrc hs;
    int32_t nt[4];
    rc = pmx_binplan_halo_source(plan, canvas, elsize, &hs.halo, &hs.S, nt, last);
    if (rc) return rc;
    hs.nt0 = nt[0]; hs.nt1 = nt[1]; hs.nt2 = nt[2];
    // planes are counted in tile space: a slab rank's block starts nt[3] = S - 1 planes into it (plane 0 of tile
    // layer 0 lies below the block and nothing wraps along that axis: the first planes of the block are then never
    // the first of a tile layer, so the gather never looks for a layer before the first)
    hs.x0 = (int)x0 + nt[3];
    PMX_REQUIRE(nrows % rows_per_plane == 0, PMX_EINVAL, "whole planes only");
    PMX_REQUIRE(rows_per_plane == (int64_t)nt[1] * T1 && n == (int64_t)nt[2] * T2 && x0 >= 0 &&
                (x0 + nt[3]) * rows_per_plane + nrows <= (int64_t)nt[0] * T0 * rows_per_plane, PMX_EINVAL,
                "rows do not match the mesh the plan's tiles cover");
    const int64_t plane_extra = plane_pitch - rows_per_plane * pitch;
    if (nrows == 0) return PMX_OK;
    hipStream_t st = (hipStream_t)stream;
    void *tw = nullptr;
    rc = get_twiddles((int)n, elsize, &tw, st);
    if (rc) return rc;
    int logm = length_code(n / 2);
    if (elsize == 8)
        return dispatch_logm<double>(logm, data, nrows, pitch, scale, tw, false, rows_per_plane, plane_extra, st, &hs, dst, nullptr);
    return dispatch_logm<float>(logm, data, nrows, pitch, scale, tw, false, rows_per_plane, plane_extra, st, &hs, dst, nullptr);
}

// [r6] PMX_OK if pmx_rowfft_split is built for rows of n reals cut into nparts ranges: every length of pmx_rowfft,
// at most PMX_MAXSEG ranges
extern "C" int pmx_rowfft_split_supported(int64_t n, int32_t elsize, int32_t nparts)
{
    int rc = pmx_rowfft_supported(n, elsize);
    if (rc) return rc;
    return (nparts >= 1 && nparts <= PMX_MAXSEG) ? PMX_OK : PMX_EUNSUPPORTED;
}

// The row pass of a pencil transform with the last-axis split of its first global transpose on it (RowSeg above):
// inverse = 0: src = nrows rows of n reals (row pitch `pitch` complex elements) -> dst = the n/2 + 1 modes of every row
// in nparts blocks, block q = the modes [offsets[q], offsets[q + 1]) of all rows, dense, at element nrows * offsets[q];
// inverse = 1: src = those blocks -> dst = rows of n reals.  Out of place.  What pmx_rowfft + pmx_slab_pack (n0 = nrows,
// n1 = n/2 + 1, n2 = 1) make in two sweeps.
extern "C" int pmx_rowfft_split(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t nrows, int64_t n,
                                int64_t pitch, double scale, const int64_t *offsets, int32_t nparts, void *stream)
{
    int rc = pmx_rowfft_split_supported(n, elsize, nparts);
    if (rc) { set_error("pmx_rowfft_split: rows of %lld reals in %d ranges are not built", (long long)n, (int)nparts); return rc; }
    PMX_REQUIRE(src != nullptr && dst != nullptr && src != dst && nrows >= 0 && pitch >= n / 2 + 1, PMX_EINVAL, "bad arguments");
    PMX_REQUIRE(offsets != nullptr && offsets[0] == 0 && offsets[nparts] == n / 2 + 1, PMX_EINVAL,
                "offsets must run from 0 to n/2 + 1");
    // (the kernel indexes rows of the blocks with 32-bit widths times 64-bit rows: nothing to bound here)
    RowSeg seg;
    seg.n = nparts;
    for (int q = 0; q <= PMX_MAXSEG; q++) seg.e[q] = (int)(n / 2 + 1);
    for (int q = 0; q <= nparts; q++) {
        PMX_REQUIRE(q == 0 || offsets[q] >= offsets[q - 1], PMX_EINVAL, "offsets must not decrease");
        seg.e[q] = (int)offsets[q];
    }
    // the range that holds the Nyquist mode: the last one that is not empty
    seg.last = 0;
    for (int q = 0; q < nparts; q++)
        if (offsets[q + 1] > offsets[q]) seg.last = (int)offsets[q];
    if (nrows == 0) return PMX_OK;
    hipStream_t st = (hipStream_t)stream;
    void *tw = nullptr;
    rc = get_twiddles((int)n, elsize, &tw, st);
    if (rc) return rc;
    int logm = length_code(n / 2);
    if (elsize == 8)
        return dispatch_logm<double>(logm, (void *)src, nrows, pitch, scale, tw, inverse != 0, 0, 0, st, nullptr, dst, &seg);
    return dispatch_logm<float>(logm, (void *)src, nrows, pitch, scale, tw, inverse != 0, 0, 0, st, nullptr, dst, &seg);
}
#endif

static ColAddr plain_addr(int64_t N, int64_t B)
{
    ColAddr a;
    a.sa = N * B; a.shi = 0; a.sn = B; a.sh = 31; a.mask = 0x7fffffff; a.cw = 0; a.cpitch = 0;
    return a;
}

// PMX_OK if a column FFT of length n (element size elsize = 4|8 per component) is built
#if PMX_COLFFT_PART != 2
extern "C" int pmx_colfft_supported(int64_t n, int32_t elsize)
{
    if (elsize != 4 && elsize != 8) return PMX_EINVAL;
    int lc = length_code(n);
    if (lc < 0) return PMX_EUNSUPPORTED;
    if (lc < 16) return (n >= 64 && n <= 2048) ? PMX_OK : PMX_EUNSUPPORTED;
    if (lc < 32) {
        if (n < 192 || n > 1536) return PMX_EUNSUPPORTED;                // 3 * 2^k: 192 ... 1536
        return (n == 1536 && elsize == 4) ? PMX_EUNSUPPORTED : PMX_OK;   // (float 1536 would need 1536 threads)
    }
    if (n < 320 || n > 1280) return PMX_EUNSUPPORTED;                    // 5 * 2^k: 320, 640, 1280
    return (n == 1280 && elsize == 4) ? PMX_EUNSUPPORTED : PMX_OK;
}
#endif

// In-place FFT along the middle axis of the (A, N, B) complex array `data`.
// inverse: 0 forward (exp(-i..)), 1 backward; result multiplied by `scale`.
// t == NULL: plain transform.  t != NULL (SIMPLE transfers only: no gauss/deconv, spectral
// gradient, laplace_pow in -1..1): A must be 1 and B = n1*n2; element (i0, i1, i2) is
// multiplied by T(k) before the transform, with the index bookkeeping of pmx_apply_transfer.
#if PMX_COLFFT_PART != 2
static int colfft_any(int32_t elsize, int32_t inverse, const void *src, void *data, int64_t A, int64_t N, int64_t B,
                      double scale, const pmx_transfer *t, int64_t n1, int64_t n2, const int64_t *start,
                      const int64_t *nmesh, const double *boxsize, int64_t a_stride, int64_t n_stride,
                      void *stream)
{
    int rc = pmx_colfft_supported(N, elsize);
    if (rc) { set_error("pmx_colfft: unsupported length %lld", (long long)N); return rc; }
    PMX_REQUIRE(data != nullptr && src != nullptr && A >= 0 && B >= 0, PMX_EINVAL, "bad arguments");
    if (A == 0 || B == 0) return PMX_OK;
    ColGeom g;
    g.A = A; g.B = B; g.N = (int32_t)N; g.scale = scale;
    g.logN = length_code(N);
    g.n1 = 1; g.n2 = 1;
    g.in = g.out = plain_addr(N, B);
    // padded layouts: a_stride = elements between successive a, n_stride between successive n
    PMX_REQUIRE(a_stride == 0 || a_stride >= N * B, PMX_EINVAL, "a_stride smaller than a dense (N, B) block");
    PMX_REQUIRE(n_stride == 0 || n_stride >= B, PMX_EINVAL, "n_stride smaller than B");
    PMX_REQUIRE(n_stride == 0 || n_stride == B || A == 1, PMX_EINVAL, "a padded line stride needs A == 1");
    if (a_stride) g.in.sa = g.out.sa = a_stride;
    if (n_stride) g.in.sn = g.out.sn = n_stride;
    bool apply = t != nullptr;
    if (apply) {
        PMX_REQUIRE(A == 1 && n1 * n2 == B && B < (1ll << 31), PMX_EINVAL,
                    "fused transfer needs the axis-0 pass of one block");
        PMX_REQUIRE(t->gauss_r == 0 && t->deconv_pow == 0 && (t->grad_dir < 0 || t->grad_kind == 0 || (t->grad_kind == 1 && t->grad_dir > 0)) &&
                    t->laplace_pow >= -1 && t->laplace_pow <= 1 && t->grad_dir < 3,
                    PMX_EUNSUPPORTED, "only the closed-form transfers without transcendentals can be fused");
        g.t = *t;
        rc = gradient_table(g, t, nmesh, boxsize);
        if (rc) return rc;
        g.n1 = (int32_t)n1; g.n2 = (int32_t)n2;
        for (int d = 0; d < 3; d++) {
            g.start[d] = start[d]; g.nmesh[d] = nmesh[d];
            g.dw[d] = 2 * M_PI / nmesh[d];
            g.nl[d] = nmesh[d] / boxsize[d];
        }
    }
    hipStream_t st = (hipStream_t)stream;
    void *tw = nullptr;
    rc = get_twiddles((int)N, elsize, &tw, st);
    if (rc) return rc;
    if (elsize == 8) return dispatch_logn<double>(g, src, data, tw, inverse != 0, apply, st);
    // float: 16 columns x 8 B = 128-byte rows, 16 lines per thread: 1024 threads at N = 1024
    return dispatch_logn<float>(g, src, data, tw, inverse != 0, apply, st);
}

extern "C" int pmx_colfft(int32_t elsize, int32_t inverse, void *data, int64_t A, int64_t N, int64_t B,
                          double scale, const pmx_transfer *t, int64_t n1, int64_t n2, const int64_t *start,
                          const int64_t *nmesh, const double *boxsize, int64_t a_stride, int64_t n_stride,
                          void *stream)
{
    return colfft_any(elsize, inverse, data, data, A, N, B, scale, t, n1, n2, start, nmesh, boxsize, a_stride, n_stride, stream);
}

// the same pass from `src` into `dst` (same layout, no overlap unless equal): the first pass of a transform whose
// caller keeps its input (c2r() / r2c() with out=None, the reference's default: pm.py:655-694, 987-1019) reads the
// input and writes the work buffer, instead of a copy of the whole array in front of an in-place pass
extern "C" int pmx_colfft_to(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t A, int64_t N, int64_t B,
                             double scale, const pmx_transfer *t, int64_t n1, int64_t n2, const int64_t *start,
                             const int64_t *nmesh, const double *boxsize, int64_t a_stride, int64_t n_stride,
                             void *stream)
{
    return colfft_any(elsize, inverse, src, dst, A, N, B, scale, t, n1, n2, start, nmesh, boxsize, a_stride, n_stride, stream);
}
#endif

// Tile width of the round-trip kernel: where the 128-byte tile leaves room for one workgroup per CU only, 64-byte
// rows (two or three workgroups).  Unlike the plain passes — which are at the rate of their bare loads and stores
// either way — this kernel computes for a third of its time (two transforms and the transfer function), and a
// second workgroup on the CU is what overlaps that with memory: 1024^3 in double 6.8 -> 5.35 ms per launch,
// c2r of 1024^3 in float 8.4 -> 7.7 ms, of 768^3 in double 6.8 -> 6.7 ms.  (N = 640 in double measured the other way,
// 3.9 -> 4.2 ms, and keeps its 128-byte rows; at N = 512, where two workgroups of the 128-byte tile share a CU already,
// 64-byte rows cost fp64 2 % and gave fp32 2 %: PMX_ROUND_NARROW_FROM stays at the one-workgroup limit.)
#ifndef PMX_ROUND_NARROW
#define PMX_ROUND_NARROW 1
#endif
#ifndef PMX_ROUND_NARROW_FROM
#define PMX_ROUND_NARROW_FROM (80 * 1024)
#endif
template <typename T, int LOGN, int RB0> struct RoundRB {
    static constexpr int value = (PMX_ROUND_NARROW && RB0 > 64 && (LOGN < 16 || LOGN == 24) && ColPipe<T, LOGN, RB0>::bytes > PMX_ROUND_NARROW_FROM) ? 64 : RB0;
};
template <typename T, int LOGN, int RB>
static int launch_round(const ColGeom &g, void *data, const void *tw, bool apply, hipStream_t st)
{
    constexpr int N = Len<LOGN>::N;
    constexpr int W = RB / (int)sizeof(cpx<T>);
    constexpr int NT = N / Rpt<T, LOGN>::value * W;
    static_assert(NT <= 1024 && NT % 64 == 0, "round-trip kernel: workgroup size");
    size_t lds = (size_t)(N * W + (HalfTw<T, LOGN, RB>::value ? N / 2 : N)) * sizeof(cpx<T>);
    int64_t tiles = (g.B + W - 1) / W;
    PMX_REQUIRE(tiles < (1ll << 31), PMX_EUNSUPPORTED, "more than 2^31 tiles in one column pass");
    unsigned grid = (unsigned)tiles;
    if (RoundPipe<T, LOGN, RB>::value && (!RoundPipe2<T, LOGN, RB>::value || g_persistent.load(std::memory_order_relaxed))) {
        // (the two-per-CU form follows pmx_colfft_configure like the column passes: one tile per workgroup on several ranks)
        const int64_t cus = compute_units() * (RoundPipe2<T, LOGN, RB>::value ? 2 : 1);
        grid = (unsigned)(tiles < cus ? tiles : cus);
    }
    ColGeom gx = g;
    gx.xcd = PMX_COL_XCD >= 1;
#define LAUNCH(AP)                                                                                             \
    do {                                                                                                       \
        auto k = colfft_round_kernel<T, LOGN, AP, RB>;                                                         \
        PMX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k<<<grid, NT, lds, st>>>(gx, (cpx<T> *)data, (const cpx<T> *)tw);                                      \
    } while (0)
    if (apply) LAUNCH(true); else LAUNCH(false);
#undef LAUNCH
    PMX_HIP_CHECK(hipGetLastError());
    return PMX_OK;
}

template <typename T>
PMX_DISPATCH int dispatch_round(const ColGeom &g, void *data, const void *tw, bool apply, hipStream_t st)
{
    switch (g.logN) {
#define ROUND(LC, RB0) return launch_round<T, LC, RoundRB<T, LC, (RB0)>::value>(g, data, tw, apply, st)
    case 6: ROUND(6, 128);
    case 7: ROUND(7, 128);
    case 8: ROUND(8, 128);
    case 9: ROUND(9, 128);
    case 10: ROUND(10, 128);
    case 11: ROUND(11, 64);
    case 22: ROUND(22, 128);
    case 23: ROUND(23, 128);
    case 24: ROUND(24, PMX_RB768);
    case 38: ROUND(38, 128);
    case 39: ROUND(39, (sizeof(T) == 4 ? 64 : 128));
#undef ROUND
    }
    set_error("pmx_colfft_roundtrip: length code %d is not built", g.logN);
    return PMX_EUNSUPPORTED;
}
#if PMX_COLFFT_PART == 1
extern template int dispatch_round<float>(const ColGeom &, void *, const void *, bool, hipStream_t);
#elif PMX_COLFFT_PART == 2
template int dispatch_round<float>(const ColGeom &, void *, const void *, bool, hipStream_t);
#endif

#if PMX_COLFFT_PART != 2
extern "C" int pmx_colfft_configure(int32_t persistent)
{
    g_persistent.store(persistent ? 1 : 0, std::memory_order_relaxed);
    return PMX_OK;
}
#endif

#if PMX_COLFFT_PART != 2
extern "C" int pmx_colfft_roundtrip_supported(int64_t n, int32_t elsize)
{
    int rc = pmx_colfft_supported(n, elsize);
    if (rc) return rc;
    const int code = length_code(n);
    return (code == 25 || code == 40) ? PMX_EUNSUPPORTED : PMX_OK;      // 1536 / 1280: one-precision kernels, not built here
}
#endif

#if PMX_COLFFT_PART != 2
extern "C" int pmx_colfft_roundtrip(int32_t elsize, void *data, int64_t N, int64_t B, double scale,
                                    const pmx_transfer *t, int64_t n1, int64_t n2, const int64_t *start,
                                    const int64_t *nmesh, const double *boxsize, int64_t n_stride, void *stream)
{
    int rc = pmx_colfft_roundtrip_supported(N, elsize);
    if (rc) { set_error("pmx_colfft_roundtrip: unsupported length %lld", (long long)N); return rc; }
    PMX_REQUIRE(data != nullptr && B >= 0, PMX_EINVAL, "bad arguments");
    if (B == 0) return PMX_OK;
    ColGeom g;
    g.A = 1; g.B = B; g.N = (int32_t)N; g.scale = scale;
    g.logN = length_code(N);
    g.n1 = 1; g.n2 = 1;
    g.in = g.out = plain_addr(N, B);
    PMX_REQUIRE(n_stride == 0 || n_stride >= B, PMX_EINVAL, "n_stride smaller than B");
    if (n_stride) g.in.sn = g.out.sn = n_stride;
    const bool apply = t != nullptr;
    if (apply) {
        PMX_REQUIRE(n1 * n2 == B && B < (1ll << 31), PMX_EINVAL, "fused transfer needs the axis-0 pass of one block");
        PMX_REQUIRE(t->gauss_r == 0 && t->deconv_pow == 0 && (t->grad_dir < 0 || t->grad_kind == 0 || (t->grad_kind == 1 && t->grad_dir > 0)) &&
                    t->laplace_pow >= -1 && t->laplace_pow <= 1 && t->grad_dir < 3,
                    PMX_EUNSUPPORTED, "only the closed-form transfers without transcendentals can be fused");
        g.t = *t;
        rc = gradient_table(g, t, nmesh, boxsize);
        if (rc) return rc;
        g.n1 = (int32_t)n1; g.n2 = (int32_t)n2;
        for (int d = 0; d < 3; d++) {
            g.start[d] = start[d]; g.nmesh[d] = nmesh[d];
            g.dw[d] = 2 * M_PI / nmesh[d];
            g.nl[d] = nmesh[d] / boxsize[d];
        }
    }
    hipStream_t st = (hipStream_t)stream;
    void *tw = nullptr;
    rc = get_twiddles((int)N, elsize, &tw, st);
    if (rc) return rc;
    if (elsize == 8) return dispatch_round<double>(g, data, tw, apply, st);
    return dispatch_round<float>(g, data, tw, apply, st);
}
#endif

// The axis-1 pass of a slab-decomposed transform fused with the pack / unpack around the
// global transpose (the work of pmx_slab_pack for equal, power-of-two ranges of nsplit lines):
//   inverse = 0: src is the plain (A, N, B) array; dst receives the transform in split
//                layout, block r = lines [r*nsplit, (r+1)*nsplit) as a contiguous (A, nsplit, B)
//                array — the send buffer of the all-to-all;
//   inverse = 1: src is that split layout (the receive buffer), dst the plain (A, N, B) array.
// src and dst must not overlap.
#if PMX_COLFFT_PART != 2
extern "C" int pmx_colfft_split(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t A,
                                int64_t N, int64_t B, int64_t nsplit, double scale, int64_t plain_pitch,
                                void *stream)
{
    int rc = pmx_colfft_supported(N, elsize);
    if (rc) { set_error("pmx_colfft_split: unsupported length %lld", (long long)N); return rc; }
    PMX_REQUIRE(src != nullptr && dst != nullptr && src != dst && A >= 0 && B >= 0, PMX_EINVAL, "bad arguments");
    PMX_REQUIRE(nsplit >= 1 && nsplit <= N && (nsplit & (nsplit - 1)) == 0, PMX_EUNSUPPORTED,
                "nsplit must be a power of two <= N");
    if (A == 0 || B == 0) return PMX_OK;
    ColGeom g;
    g.A = A; g.B = B; g.N = (int32_t)N; g.scale = scale;
    g.logN = length_code(N);
    g.n1 = 1; g.n2 = 1;
    ColAddr split;
    split.sh = 0;
    while ((1ll << split.sh) < nsplit) split.sh++;
    split.mask = (int32_t)(nsplit - 1);
    split.sa = nsplit * B;
    split.sn = B;
    split.cw = 0; split.cpitch = 0;
    split.shi = A * nsplit * B;
    // the plain side may have padded lines (plain_pitch >= B elements per line, e.g. rows
    // rounded up to 128 bytes); the split side is always dense
    PMX_REQUIRE(plain_pitch == 0 || plain_pitch >= B, PMX_EINVAL, "plain_pitch smaller than B");
    ColAddr plain = plain_addr(N, plain_pitch ? plain_pitch : B);
    g.in = inverse ? split : plain;
    g.out = inverse ? plain : split;
    hipStream_t st = (hipStream_t)stream;
    void *tw = nullptr;
    rc = get_twiddles((int)N, elsize, &tw, st);
    if (rc) return rc;
    if (elsize == 8) return dispatch_logn<double>(g, src, dst, tw, inverse != 0, false, st);
    return dispatch_logn<float>(g, src, dst, tw, inverse != 0, false, st);
}
#endif

// The axis-1 pass of a PENCIL transform, between its two global transposes: both sides are split
// layouts of the same (A, N, B) array — src cut into ranges of nsplit_in lines (what the
// all-to-all of one process-mesh direction delivered), dst into ranges of nsplit_out lines (what
// the all-to-all of the other direction sends); 0 = plain dense.  The unpack before and the pack
// after the pass (two pmx_slab_pack sweeps over the block) ride on its load and store.
#if PMX_COLFFT_PART != 2
extern "C" int pmx_colfft_resplit(int32_t elsize, int32_t inverse, const void *src, void *dst, int64_t A,
                                  int64_t N, int64_t B, int64_t nsplit_in, int64_t nsplit_out, double scale,
                                  void *stream)
{
    int rc = pmx_colfft_supported(N, elsize);
    if (rc) { set_error("pmx_colfft_resplit: unsupported length %lld", (long long)N); return rc; }
    PMX_REQUIRE(src != nullptr && dst != nullptr && src != dst && A >= 0 && B >= 0, PMX_EINVAL, "bad arguments");
    for (int64_t ns : {nsplit_in, nsplit_out})
        PMX_REQUIRE(ns == 0 || (ns >= 1 && ns <= N && (ns & (ns - 1)) == 0 && N % ns == 0), PMX_EUNSUPPORTED,
                    "nsplit must be 0 or a power of two dividing N");
    if (A == 0 || B == 0) return PMX_OK;
    ColGeom g;
    g.A = A; g.B = B; g.N = (int32_t)N; g.scale = scale;
    g.logN = length_code(N);
    g.n1 = 1; g.n2 = 1;
    auto make = [&](int64_t nsplit) {
        if (nsplit == 0) return plain_addr(N, B);
        ColAddr a;
        a.sh = 0;
        while ((1ll << a.sh) < nsplit) a.sh++;
        a.mask = (int32_t)(nsplit - 1);
        a.sa = nsplit * B; a.sn = B; a.cw = 0; a.cpitch = 0;
        a.shi = A * nsplit * B;
        return a;
    };
    g.in = make(nsplit_in);
    g.out = make(nsplit_out);
    hipStream_t st = (hipStream_t)stream;
    void *tw = nullptr;
    rc = get_twiddles((int)N, elsize, &tw, st);
    if (rc) return rc;
    if (elsize == 8) return dispatch_logn<double>(g, src, dst, tw, inverse != 0, false, st);
    return dispatch_logn<float>(g, src, dst, tw, inverse != 0, false, st);
}
#endif

// The axis-0 pass of a slab transform on ONE chunk of the last axis (pipelined transposes: the
// all-to-all of chunk c overlaps the passes of chunks c-1 and c+1).  `full` is the standard
// (N, n1, pitch) block of the transposed complex field; the chunk is its columns
// [coff, coff + cw) of every n1 row, and `chunk` is the dense (N, n1, cw) buffer that the
// all-to-all delivers (r2c: what the ranks sent, row-major by source = line index) or takes.
//   to_full = 1: FFT along N of `chunk`, result scattered into `full` (r2c, last stage);
//   to_full = 0: FFT along N of the chunk's columns gathered from `full`, result dense in
//                `chunk` (c2r, first stage; t != NULL multiplies by the transfer function first,
//                see pmx_colfft; start[] is the global start of the full block).
#if PMX_COLFFT_PART != 2
extern "C" int pmx_colfft_chunk(int32_t elsize, int32_t inverse, void *chunk, void *full, int64_t N,
                                int64_t n1, int64_t cw, int64_t pitch, int64_t coff, int32_t to_full,
                                double scale, const pmx_transfer *t, const int64_t *start,
                                const int64_t *nmesh, const double *boxsize, void *stream)
{
    int rc = pmx_colfft_supported(N, elsize);
    if (rc) { set_error("pmx_colfft_chunk: unsupported length %lld", (long long)N); return rc; }
    PMX_REQUIRE(chunk != nullptr && full != nullptr && chunk != full, PMX_EINVAL, "bad buffers");
    PMX_REQUIRE(n1 >= 0 && cw >= 0 && coff >= 0 && coff + cw <= pitch, PMX_EINVAL, "bad chunk geometry");
    if (n1 == 0 || cw == 0) return PMX_OK;
    const int64_t B = n1 * cw;
    ColGeom g;
    g.A = 1; g.B = B; g.N = (int32_t)N; g.scale = scale;
    g.logN = length_code(N);
    g.n1 = 1; g.n2 = 1;
    ColAddr dense = plain_addr(N, B);
    ColAddr mapped = plain_addr(N, n1 * pitch);
    mapped.cw = cw;
    mapped.cpitch = pitch;
    g.in = to_full ? dense : mapped;
    g.out = to_full ? mapped : dense;
    bool apply = t != nullptr;
    if (apply) {
        PMX_REQUIRE(!to_full && B < (1ll << 31), PMX_EINVAL, "the fused transfer belongs to the gathering pass");
        PMX_REQUIRE(t->gauss_r == 0 && t->deconv_pow == 0 && (t->grad_dir < 0 || t->grad_kind == 0 || (t->grad_kind == 1 && t->grad_dir > 0)) &&
                    t->laplace_pow >= -1 && t->laplace_pow <= 1 && t->grad_dir < 3,
                    PMX_EUNSUPPORTED, "only the closed-form transfers without transcendentals can be fused");
        g.t = *t;
        rc = gradient_table(g, t, nmesh, boxsize);
        if (rc) return rc;
        g.n1 = (int32_t)n1; g.n2 = (int32_t)cw;        // column b = (i1, i2 - coff)
        for (int d = 0; d < 3; d++) {
            g.start[d] = start[d] + (d == 2 ? coff : 0);
            g.nmesh[d] = nmesh[d];
            g.dw[d] = 2 * M_PI / nmesh[d];
            g.nl[d] = nmesh[d] / boxsize[d];
        }
    }
    size_t es = 2 * (size_t)elsize;
    const void *src = to_full ? chunk : (const void *)((const char *)full + coff * es);
    void *dst = to_full ? (void *)((char *)full + coff * es) : chunk;
    hipStream_t st = (hipStream_t)stream;
    void *tw = nullptr;
    rc = get_twiddles((int)N, elsize, &tw, st);
    if (rc) return rc;
    if (elsize == 8) return dispatch_logn<double>(g, src, dst, tw, inverse != 0, apply, st);
    return dispatch_logn<float>(g, src, dst, tw, inverse != 0, apply, st);
}
#endif
