// catan_nn.hip - fused small-sequence multi-head attention (forward + backward) for the policy net.
//
// The reference net attends over 19 hex tiles (4 heads x 16) and over <= 25 development cards (4 heads x 4)
// (RL/models/tile_encoder.py:41-60, player_modules.py:55-69, multi_headed_attention.py:25-54).  As library batched GEMMs
// these are 65 536 x 4 products of 19x16 matrices - 55-60 % of the net's time on MI355X (rocprof/torch profiler).
// Here one wave handles G = 64 / L sequences: lane = (sequence, query row); K and V of the G sequences sit in LDS
// (fp32), scores / softmax / PV stay in registers.  The arithmetic is a few kFLOP per sequence, so the kernels are
// HBM-bound on the qkv read and the output write; plain VALU FMAs (no MFMA: the tiles are 19x16).
// Layout: qkv [B][L][3][H][HD] (the fused QKV projection output), out [B][L][H*HD]; T = float or bf16 storage, fp32 math.
// lens (optional): keys j >= lens[b] are masked (the reference's key mask); rows are computed for every query position
// (the caller zeroes padded rows, as the reference does).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

namespace catan {

template <class T> __device__ __forceinline__ float ld_f(const T* p);
template <> __device__ __forceinline__ float ld_f<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f<__hip_bfloat16>(const __hip_bfloat16* p) { return __bfloat162float(*p); }
template <class T> __device__ __forceinline__ void st_f(T* p, float v);
template <> __device__ __forceinline__ void st_f<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f<__hip_bfloat16>(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

// two floats -> one register of two bf16 (a in the low half), round to nearest even: ONE v_cvt_pk_bf16_f32.  (Written per element -
// f2bf(a) | f2bf(b) << 16 - the compiler pairs the conversions of elements (0, 2) and (1, 3) of a group of four and re-interleaves
// the halves with and / shift / two or_sdwa: six instructions per four elements instead of two.)
typedef __bf16 bf16pair_t __attribute__((ext_vector_type(2)));
typedef float f32pair_t __attribute__((ext_vector_type(2)));
// CATAN_PK_BF_SCALAR (a build macro, off): the per-element form, kept as the fall-back should the packed conversion ever have to go
// (the run-to-run instability of round 4 appeared only next to packed-f32 VALU, which the build now excludes and checks for;
// tools/stress_determinism.py covers every kernel that converts through this function).
__device__ __forceinline__ unsigned pk_bf(float a, float b) {
#ifdef CATAN_PK_BF_SCALAR
    const __hip_bfloat16 ha = __float2bfloat16(a), hb = __float2bfloat16(b);
    return (unsigned)*reinterpret_cast<const unsigned short*>(&ha) | ((unsigned)*reinterpret_cast<const unsigned short*>(&hb) << 16);
#else
    const f32pair_t f = {a, b};
    union { bf16pair_t h; unsigned u; } r;
    r.h = __builtin_convertvector(f, bf16pair_t);
    return r.u;
#endif
}

// LDS rows are kept in the storage type (bf16 inputs: half the LDS, twice the blocks per CU) with a 16 B aligned pitch; a head
// slice (HD = 4 or 16 consecutive elements) is fetched with 8 / 16 B LDS reads and widened in registers.
template <class T> struct LdsRow;
template <> struct LdsRow<float> {
    static constexpr int PAD = 4;
    template <int HD> static __device__ __forceinline__ void load(const float* p, float (&v)[HD]) {
#pragma unroll
        for (int d = 0; d < HD; d += 4) { const float4 u = *reinterpret_cast<const float4*>(p + d); v[d] = u.x; v[d + 1] = u.y; v[d + 2] = u.z; v[d + 3] = u.w; }
    }
};
template <> struct LdsRow<__hip_bfloat16> {
    static constexpr int PAD = 8;
    template <int HD> static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[HD]) {
        if constexpr (HD == 4) {
            const uint2 u = *reinterpret_cast<const uint2*>(p);
            v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xFFFF0000u);
            v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xFFFF0000u);
        } else {
#pragma unroll
            for (int d = 0; d < HD; d += 8) {
                const uint4 u = *reinterpret_cast<const uint4*>(p + d);
                const unsigned w[4] = { u.x, u.y, u.z, u.w };
#pragma unroll
                for (int k = 0; k < 4; k++) { v[d + 2 * k] = __uint_as_float(w[k] << 16); v[d + 2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u); }
            }
        }
    }
};

template <class T, int L, int H, int HD>
__global__ __launch_bounds__(64) void k_attn_fwd(const T* __restrict__ qkv, const int* __restrict__ lens, T* __restrict__ out, long B) {
    constexpr int G = 64 / L, D = H * HD;
    constexpr int P = D + LdsRow<float>::PAD;              // forward: fp32 rows (half-width rows measured slower here)
    __shared__ __attribute__((aligned(16))) float Ks[G][L][P], Vs[G][L][P];
    const int lane = threadIdx.x, g = lane / L, i = lane % L;
    const long b0 = (long)blockIdx.x * G;
    // stage K, V (coalesced over the contiguous [L][3][D] block of each sequence)
    for (int gg = 0; gg < G; gg++) {
        const long b = b0 + gg;
        if (b >= B) break;
        const T* base = qkv + b * (long)(L * 3 * D);
        for (int x = lane; x < L * D; x += 64) {
            const int l = x / D, d = x % D;
            Ks[gg][l][d] = ld_f(base + (l * 3 + 1) * D + d);
            Vs[gg][l][d] = ld_f(base + (l * 3 + 2) * D + d);
        }
    }
    __syncthreads();
    const long b = b0 + g;
    if (g >= G || b >= B) return;
    const int len = lens ? lens[b] : L;
    const T* qp = qkv + (b * L + i) * (long)(3 * D);
    const float scale = rsqrtf((float)HD);
    T* op = out + (b * L + i) * (long)D;
    for (int h = 0; h < H; h++) {
        float q[HD];
#pragma unroll
        for (int d = 0; d < HD; d++) q[d] = ld_f(qp + h * HD + d) * scale;
        float s[L], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < L; j++) {
            float kr[HD];
            LdsRow<float>::template load<HD>(&Ks[g][j][h * HD], kr);
            float a = 0.0f;
#pragma unroll
            for (int d = 0; d < HD; d++) a += q[d] * kr[d];
            s[j] = j < len ? a : -INFINITY;
            mx = fmaxf(mx, s[j]);
        }
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < L; j++) { s[j] = __expf(s[j] - mx); sum += s[j]; }
        const float inv = 1.0f / sum;
        float o[HD];
#pragma unroll
        for (int d = 0; d < HD; d++) o[d] = 0.0f;
#pragma unroll
        for (int j = 0; j < L; j++) {
            const float p = s[j] * inv;
            float vr[HD];
            LdsRow<float>::template load<HD>(&Vs[g][j][h * HD], vr);
#pragma unroll
            for (int d = 0; d < HD; d++) o[d] += p * vr[d];
        }
#pragma unroll
        for (int d = 0; d < HD; d++) st_f(op + h * HD + d, o[d]);
    }
}

// dqkv [B][L][3][H][HD] from dout [B][L][D]; probabilities are recomputed (no saved attention matrix)
template <class T, int L, int H, int HD>
__global__ __launch_bounds__(64) void k_attn_bwd(const T* __restrict__ qkv, const int* __restrict__ lens, const T* __restrict__ dout,
                                                 T* __restrict__ dqkv, long B) {
    constexpr int G = 64 / L, D = H * HD;
    constexpr int P = D + LdsRow<T>::PAD;
    __shared__ __attribute__((aligned(16))) T Qs[G][L][P], Ks[G][L][P], Vs[G][L][P], Os[G][L][P];
    __shared__ float Ps[G][L][L + 1], Ss[G][L][L + 1];
    const int lane = threadIdx.x, g = lane / L, i = lane % L;
    const long b0 = (long)blockIdx.x * G;
    for (int gg = 0; gg < G; gg++) {
        const long b = b0 + gg;
        if (b >= B) break;
        const T* base = qkv + b * (long)(L * 3 * D);
        const T* dob = dout + b * (long)(L * D);
        for (int x = lane; x < L * D; x += 64) {
            const int l = x / D, d = x % D;
            Qs[gg][l][d] = base[(l * 3 + 0) * D + d];
            Ks[gg][l][d] = base[(l * 3 + 1) * D + d];
            Vs[gg][l][d] = base[(l * 3 + 2) * D + d];
            Os[gg][l][d] = dob[l * D + d];
        }
    }
    __syncthreads();
    const long b = b0 + g;
    const bool act = g < G && b < B;
    const int len = act ? (lens ? lens[b] : L) : 0;
    const float scale = rsqrtf((float)HD);
    T* dp = dqkv + (b * L + i) * (long)(3 * D);
    for (int h = 0; h < H; h++) {
        if (act) {
            float s[L], mx = -INFINITY;
            float qi[HD], oi[HD];
            LdsRow<T>::template load<HD>(&Qs[g][i][h * HD], qi);
            LdsRow<T>::template load<HD>(&Os[g][i][h * HD], oi);
#pragma unroll
            for (int j = 0; j < L; j++) {
                float kr[HD];
                LdsRow<T>::template load<HD>(&Ks[g][j][h * HD], kr);
                float a = 0.0f;
#pragma unroll
                for (int d = 0; d < HD; d++) a += qi[d] * kr[d];
                s[j] = j < len ? a * scale : -INFINITY;
                mx = fmaxf(mx, s[j]);
            }
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < L; j++) { s[j] = __expf(s[j] - mx); sum += s[j]; }
            const float inv = 1.0f / sum;
            float dP[L], delta = 0.0f;
#pragma unroll
            for (int j = 0; j < L; j++) {
                s[j] *= inv;
                float vr[HD];
                LdsRow<T>::template load<HD>(&Vs[g][j][h * HD], vr);
                float a = 0.0f;
#pragma unroll
                for (int d = 0; d < HD; d++) a += oi[d] * vr[d];
                dP[j] = a;
                delta += s[j] * a;
            }
            float dq[HD];
#pragma unroll
            for (int d = 0; d < HD; d++) dq[d] = 0.0f;
#pragma unroll
            for (int j = 0; j < L; j++) {
                const float ds = s[j] * (dP[j] - delta) * scale;       // d(scores)/ ... includes the 1/sqrt(hd)
                Ps[g][i][j] = s[j];
                Ss[g][i][j] = ds;
                float kr[HD];
                LdsRow<T>::template load<HD>(&Ks[g][j][h * HD], kr);
#pragma unroll
                for (int d = 0; d < HD; d++) dq[d] += ds * kr[d];
            }
#pragma unroll
            for (int d = 0; d < HD; d++) st_f(dp + 0 * D + h * HD + d, dq[d]);
        }
        __syncthreads();
        if (act) {                                   // lane (g, j = i): column sums
            float dk[HD], dv[HD];
#pragma unroll
            for (int d = 0; d < HD; d++) { dk[d] = 0.0f; dv[d] = 0.0f; }
#pragma unroll
            for (int r = 0; r < L; r++) {
                const float ds = Ss[g][r][i], p = Ps[g][r][i];
                float qr[HD], orow[HD];
                LdsRow<T>::template load<HD>(&Qs[g][r][h * HD], qr);
                LdsRow<T>::template load<HD>(&Os[g][r][h * HD], orow);
#pragma unroll
                for (int d = 0; d < HD; d++) { dk[d] += ds * qr[d]; dv[d] += p * orow[d]; }
            }
#pragma unroll
            for (int d = 0; d < HD; d++) { st_f(dp + 1 * D + h * HD + d, dk[d]); st_f(dp + 2 * D + h * HD + d, dv[d]); }
        }
        __syncthreads();
    }
}

}  // namespace catan

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm over a small last dimension D <= 64 (+ optional fused ReLU), forward and backward.
// The net normalises millions of short rows (19 tiles x 64, 25 cards x 16, 19 x 25 per game); a library LayerNorm
// spends a workgroup per row.  Here GL = 16 / 32 / 64 lanes own one row at a time (element j of the row in lane j of the
// group: fully coalesced, group-local shuffles for the moments) and every lane always works on the same column, so the
// weight/bias gradients accumulate in registers and leave with one atomicAdd per lane per kernel.
namespace catan {

template <int GL>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = GL / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, GL);
    return v;
}

template <class T, int D, int GL>
__global__ __launch_bounds__(256) void k_ln_fwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bvec,
                                                T* __restrict__ y, long rows, float eps, int relu) {
    constexpr int RPB = 256 / GL;
    const int j = threadIdx.x % GL, r = threadIdx.x / GL;
    const bool col = j < D;
    const float wj = col ? w[j] : 0.0f, bj = col ? bvec[j] : 0.0f;
    for (long row = (long)blockIdx.x * RPB + r; row < rows; row += (long)gridDim.x * RPB) {
        const float v = col ? ld_f(x + row * D + j) : 0.0f;
        const float mean = group_sum<GL>(v) * (1.0f / D);
        const float c = col ? v - mean : 0.0f;
        const float rstd = rsqrtf(group_sum<GL>(c * c) * (1.0f / D) + eps);
        float o = c * rstd * wj + bj;
        if (relu) o = fmaxf(o, 0.0f);
        if (col) st_f(y + row * D + j, o);
    }
}

// dx, and dw/db accumulated with atomics into fp32 buffers (zeroed by the caller)
template <class T, int D, int GL>
__global__ __launch_bounds__(256) void k_ln_bwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bvec,
                                                const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ dw,
                                                float* __restrict__ db, long rows, float eps, int relu) {
    constexpr int RPB = 256 / GL;
    const int j = threadIdx.x % GL, r = threadIdx.x / GL;
    const bool col = j < D;
    const float wj = col ? w[j] : 0.0f, bj = col ? bvec[j] : 0.0f;
    float aw = 0.0f, ab = 0.0f;
    for (long row = (long)blockIdx.x * RPB + r; row < rows; row += (long)gridDim.x * RPB) {
        const float v = col ? ld_f(x + row * D + j) : 0.0f;
        const float mean = group_sum<GL>(v) * (1.0f / D);
        const float c = col ? v - mean : 0.0f;
        const float rstd = rsqrtf(group_sum<GL>(c * c) * (1.0f / D) + eps);
        const float xh = c * rstd;
        float g = col ? ld_f(dy + row * D + j) : 0.0f;
        if (relu && xh * wj + bj <= 0.0f) g = 0.0f;
        aw += g * xh; ab += g;
        const float gw = g * wj;
        const float m1 = group_sum<GL>(gw) * (1.0f / D);
        const float m2 = group_sum<GL>(gw * xh) * (1.0f / D);
        if (col) st_f(dx + row * D + j, rstd * (gw - m1 - xh * m2));
    }
    // block-level reduction over the RPB row groups, then one atomic per column per block
    __shared__ float sw[256], sb[256];
    sw[threadIdx.x] = aw; sb[threadIdx.x] = ab;
    __syncthreads();
    if (threadIdx.x < GL && threadIdx.x < D) {
        float tw = 0.0f, tb = 0.0f;
#pragma unroll
        for (int q = 0; q < RPB; q++) { tw += sw[q * GL + threadIdx.x]; tb += sb[q * GL + threadIdx.x]; }
        atomicAdd(dw + threadIdx.x, tw); atomicAdd(db + threadIdx.x, tb);
    }
}


// ------------------------------------------------------------------------------------------------ LayerNorm, wide rows
// D = GL * EPL: GL lanes per row, lane j owns the EPL contiguous elements j*EPL .. j*EPL+EPL-1 (one 4 / 8 / 16 B access per
// lane, fully coalesced), moments by a GL-lane butterfly.  GL = 64 for D = 128, 256, 512 (the player / head / value modules,
// RL/models/player_modules.py:26-30,114-117, action_heads_module.py, policy.py); GL = 8, EPL = 8 for D = 64 (the tile
// encoder's 1.2 M rows per pass: eight rows per wave, three shuffle steps instead of six).  In the backward every lane always
// works on the same columns, so dw / db accumulate in registers and leave through one LDS reduction + atomics per block.
template <class T, int EPL> struct RowVec;
template <int EPL> struct RowVec<float, EPL> {
    static __device__ __forceinline__ float rounded(float a) { return a; }
    static __device__ __forceinline__ void load(const float* p, float (&v)[EPL]) {
#pragma unroll
        for (int e = 0; e < EPL; e += 4) { const float4 u = *reinterpret_cast<const float4*>(p + e); v[e] = u.x; v[e + 1] = u.y; v[e + 2] = u.z; v[e + 3] = u.w; }
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[EPL]) {
#pragma unroll
        for (int e = 0; e < EPL; e += 4) *reinterpret_cast<float4*>(p + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
    }
};
template <> struct RowVec<float, 2> {
    static __device__ __forceinline__ float rounded(float a) { return a; }
    static __device__ __forceinline__ void load(const float* p, float (&v)[2]) { const float2 u = *reinterpret_cast<const float2*>(p); v[0] = u.x; v[1] = u.y; }
    static __device__ __forceinline__ void store(float* p, const float (&v)[2]) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
};
template <int EPL> struct RowVec<__hip_bfloat16, EPL> {
    static __device__ __forceinline__ float rounded(float a) { return __bfloat162float(__float2bfloat16(a)); }
    static __device__ __forceinline__ unsigned pack(float a, float b) {
        return pk_bf(a, b);
    }
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[EPL]) {
        unsigned w[EPL / 2];
        if constexpr (EPL == 2) w[0] = *reinterpret_cast<const unsigned*>(p);
        else if constexpr (EPL == 4) { const uint2 u = *reinterpret_cast<const uint2*>(p); w[0] = u.x; w[1] = u.y; }
        else { const uint4 u = *reinterpret_cast<const uint4*>(p); w[0] = u.x; w[1] = u.y; w[2] = u.z; w[3] = u.w; }
#pragma unroll
        for (int k = 0; k < EPL / 2; k++) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u); }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&v)[EPL]) {
        if constexpr (EPL == 2) *reinterpret_cast<unsigned*>(p) = pack(v[0], v[1]);
        else if constexpr (EPL == 4) *reinterpret_cast<uint2*>(p) = make_uint2(pack(v[0], v[1]), pack(v[2], v[3]));
        else *reinterpret_cast<uint4*>(p) = make_uint4(pack(v[0], v[1]), pack(v[2], v[3]), pack(v[4], v[5]), pack(v[6], v[7]));
    }
};

template <class T, int EPL, int GL>
__global__ __launch_bounds__(256) void k_lnw_fwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bvec,
                                                 T* __restrict__ y, long rows, float eps, int relu) {
    constexpr int D = GL * EPL, RPB = 256 / GL;
    const int j = threadIdx.x % GL, wv = threadIdx.x / GL;
    float wj[EPL], bj[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) { wj[e] = w[j * EPL + e]; bj[e] = bvec[j * EPL + e]; }
    for (long row = (long)blockIdx.x * RPB + wv; row < rows; row += (long)gridDim.x * RPB) {
        float v[EPL];
        RowVec<T, EPL>::load(x + row * D + j * EPL, v);
        float sum = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; e++) sum += v[e];
        const float mean = group_sum<GL>(sum) * (1.0f / D);
        float sq = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; e++) { v[e] -= mean; sq += v[e] * v[e]; }
        const float rstd = rsqrtf(group_sum<GL>(sq) * (1.0f / D) + eps);
#pragma unroll
        for (int e = 0; e < EPL; e++) { v[e] = v[e] * rstd * wj[e] + bj[e]; if (relu) v[e] = fmaxf(v[e], 0.0f); }
        RowVec<T, EPL>::store(y + row * D + j * EPL, v);
    }
}

__device__ __forceinline__ unsigned short f2bf_ln(float v) { const __hip_bfloat16 h = __float2bfloat16(v); return *reinterpret_cast<const unsigned short*>(&h); }
// LayerNorm over rows whose byte length is not a multiple of 16 (the 25-wide rows of the card-list / tile-output projections: 50 B
// in bf16), one LANE per row: a workgroup copies 256 consecutive rows - one contiguous, 16-byte aligned span - into LDS with 16-byte
// accesses, every lane normalises its row there, and the span leaves the same way.  k_ln_fwd / k_ln_bwd spend a lane per ELEMENT
// (2-byte accesses, 25 of 32 lanes busy): 3.4 M x 25 backward 444 us = 1.2 TB/s.
template <int D>
__global__ __launch_bounds__(256) void k_lnr_fwd(const unsigned short* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bvec,
                                                 unsigned short* __restrict__ y, long rows, float eps, int relu) {
    constexpr int SPAN = 256 * D;                                   // elements per workgroup pass (a multiple of 8: 16-byte pieces)
    static_assert(SPAN % 8 == 0, "whole 16-byte pieces");
    __shared__ __attribute__((aligned(16))) unsigned short sx[SPAN];
    __shared__ float sw[D], sb[D];
    if (threadIdx.x < D) { sw[threadIdx.x] = w[threadIdx.x]; sb[threadIdx.x] = bvec[threadIdx.x]; }
    const long total = rows * D;
    for (long r0 = (long)blockIdx.x * 256; r0 < rows; r0 += (long)gridDim.x * 256) {
        const long e0 = r0 * D;
        __syncthreads();
        for (int c = threadIdx.x; c < SPAN / 8; c += 256) {
            const long e = e0 + (long)c * 8;
            if (e + 8 <= total) *reinterpret_cast<uint4*>(sx + c * 8) = *reinterpret_cast<const uint4*>(x + e);
            else for (int k = 0; k < 8; k++) if (e + k < total) sx[c * 8 + k] = x[e + k];
        }
        __syncthreads();
        if (r0 + threadIdx.x < rows) {
            unsigned short* row = sx + threadIdx.x * D;
            float v[D], mean = 0.f;
#pragma unroll
            for (int i = 0; i < D; i++) { v[i] = __uint_as_float((unsigned)row[i] << 16); mean += v[i]; }
            mean *= 1.0f / D;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < D; i++) { v[i] -= mean; sq += v[i] * v[i]; }
            const float rstd = rsqrtf(sq * (1.0f / D) + eps);
#pragma unroll
            for (int i = 0; i < D; i++) {
                float o = v[i] * rstd * sw[i] + sb[i];
                if (relu) o = fmaxf(o, 0.f);
                row[i] = f2bf_ln(o);
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < SPAN / 8; c += 256) {
            const long e = e0 + (long)c * 8;
            if (e + 8 <= total) *reinterpret_cast<uint4*>(y + e) = *reinterpret_cast<const uint4*>(sx + c * 8);
            else for (int k = 0; k < 8; k++) if (e + k < total) y[e + k] = sx[c * 8 + k];
        }
    }
}
template <int D>
__global__ __launch_bounds__(256) void k_lnr_bwd(const unsigned short* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bvec,
                                                 const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx, float* __restrict__ dw,
                                                 float* __restrict__ db, long rows, float eps, int relu) {
    constexpr int SPAN = 256 * D;
    __shared__ __attribute__((aligned(16))) unsigned short sx[SPAN];
    __shared__ __attribute__((aligned(16))) unsigned short sg[SPAN];
    __shared__ float sw[D], sb[D], saw[D], sab[D];
    if (threadIdx.x < D) { sw[threadIdx.x] = w[threadIdx.x]; sb[threadIdx.x] = bvec[threadIdx.x]; saw[threadIdx.x] = 0.f; sab[threadIdx.x] = 0.f; }
    float aw[D], ab[D];
#pragma unroll
    for (int i = 0; i < D; i++) { aw[i] = 0.f; ab[i] = 0.f; }
    const long total = rows * D;
    for (long r0 = (long)blockIdx.x * 256; r0 < rows; r0 += (long)gridDim.x * 256) {
        const long e0 = r0 * D;
        __syncthreads();
        for (int c = threadIdx.x; c < SPAN / 8; c += 256) {
            const long e = e0 + (long)c * 8;
            if (e + 8 <= total) {
                *reinterpret_cast<uint4*>(sx + c * 8) = *reinterpret_cast<const uint4*>(x + e);
                *reinterpret_cast<uint4*>(sg + c * 8) = *reinterpret_cast<const uint4*>(dy + e);
            } else for (int k = 0; k < 8; k++) if (e + k < total) { sx[c * 8 + k] = x[e + k]; sg[c * 8 + k] = dy[e + k]; }
        }
        __syncthreads();
        if (r0 + threadIdx.x < rows) {
            const unsigned short* xr = sx + threadIdx.x * D;
            unsigned short* gr = sg + threadIdx.x * D;
            float v[D], g[D], mean = 0.f;
#pragma unroll
            for (int i = 0; i < D; i++) { v[i] = __uint_as_float((unsigned)xr[i] << 16); g[i] = __uint_as_float((unsigned)gr[i] << 16); mean += v[i]; }
            mean *= 1.0f / D;
            float sq = 0.f;
#pragma unroll
            for (int i = 0; i < D; i++) { v[i] -= mean; sq += v[i] * v[i]; }
            const float rstd = rsqrtf(sq * (1.0f / D) + eps);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < D; i++) {
                v[i] *= rstd;                                          // x_hat
                if (relu && v[i] * sw[i] + sb[i] <= 0.0f) g[i] = 0.0f;
                aw[i] += g[i] * v[i]; ab[i] += g[i];
                g[i] *= sw[i];
                s1 += g[i]; s2 += g[i] * v[i];
            }
            const float m1 = s1 * (1.0f / D), m2 = s2 * (1.0f / D);
#pragma unroll
            for (int i = 0; i < D; i++) gr[i] = f2bf_ln(rstd * (g[i] - m1 - v[i] * m2));
        }
        __syncthreads();
        for (int c = threadIdx.x; c < SPAN / 8; c += 256) {
            const long e = e0 + (long)c * 8;
            if (e + 8 <= total) *reinterpret_cast<uint4*>(dx + e) = *reinterpret_cast<const uint4*>(sg + c * 8);
            else for (int k = 0; k < 8; k++) if (e + k < total) dx[e + k] = sg[c * 8 + k];
        }
    }
    // weight / bias gradients: over the wave with shuffles, over the workgroup in LDS, one atomic per column and workgroup
#pragma unroll
    for (int i = 0; i < D; i++) {
        float a = aw[i], b = ab[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&saw[i], a); atomicAdd(&sab[i], b); }
    }
    __syncthreads();
    if (threadIdx.x < D) { atomicAdd(dw + threadIdx.x, saw[threadIdx.x]); atomicAdd(db + threadIdx.x, sab[threadIdx.x]); }
}

template <class T, int EPL, int GL>
__global__ __launch_bounds__(256) void k_lnw_bwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bvec,
                                                 const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ dw,
                                                 float* __restrict__ db, long rows, float eps, int relu, const T* __restrict__ dres = nullptr) {
    constexpr int D = GL * EPL, RPB = 256 / GL;
    const int j = threadIdx.x % GL, wv = threadIdx.x / GL;
    float wj[EPL], bj[EPL], aw[EPL], ab[EPL];
#pragma unroll
    for (int e = 0; e < EPL; e++) { wj[e] = w[j * EPL + e]; bj[e] = bvec[j * EPL + e]; aw[e] = 0.0f; ab[e] = 0.0f; }
    for (long row = (long)blockIdx.x * RPB + wv; row < rows; row += (long)gridDim.x * RPB) {
        float v[EPL], g[EPL];
        RowVec<T, EPL>::load(x + row * D + j * EPL, v);
        RowVec<T, EPL>::load(dy + row * D + j * EPL, g);
        float sum = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; e++) sum += v[e];
        const float mean = group_sum<GL>(sum) * (1.0f / D);
        float sq = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; e++) { v[e] -= mean; sq += v[e] * v[e]; }
        const float rstd = rsqrtf(group_sum<GL>(sq) * (1.0f / D) + eps);
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int e = 0; e < EPL; e++) {
            v[e] *= rstd;                                          // x_hat
            if (relu && v[e] * wj[e] + bj[e] <= 0.0f) g[e] = 0.0f;
            aw[e] += g[e] * v[e]; ab[e] += g[e];
            g[e] *= wj[e];
            s1 += g[e]; s2 += g[e] * v[e];
        }
        const float m1 = group_sum<GL>(s1) * (1.0f / D), m2 = group_sum<GL>(s2) * (1.0f / D);
#pragma unroll
        for (int e = 0; e < EPL; e++) v[e] = rstd * (g[e] - m1 - v[e] * m2);
        if (dres != nullptr) {                                     // the gradient of a second use of x (the residual stream of a pre-norm
            float r[EPL];                                          // sub-layer), added as autograd would add the two tensors
            RowVec<T, EPL>::load(dres + row * D + j * EPL, r);
#pragma unroll
            for (int e = 0; e < EPL; e++) v[e] = RowVec<T, EPL>::rounded(v[e]) + r[e];
        }
        RowVec<T, EPL>::store(dx + row * D + j * EPL, v);
    }
    __shared__ float sw[RPB][D + 1], sb[RPB][D + 1];
#pragma unroll
    for (int e = 0; e < EPL; e++) { sw[wv][j * EPL + e] = aw[e]; sb[wv][j * EPL + e] = ab[e]; }
    __syncthreads();
    for (int cidx = threadIdx.x; cidx < D; cidx += 256) {
        float tw = 0.0f, tb = 0.0f;
#pragma unroll
        for (int q = 0; q < RPB; q++) { tw += sw[q][cidx]; tb += sb[q][cidx]; }
        atomicAdd(dw + cidx, tw); atomicAdd(db + cidx, tb);
    }
}

// ------------------------------------------------------------------------------------------------ tall-skinny weight gradient
// dW[o][i] = sum_r dY[r][o] * X[r][i],  db[o] = sum_r dY[r][o]   for the Linear layers of the net whose row count is
// huge and whose widths are small (tile encoder: R = 19 B rows, widths 25..192; card attention: R = 25 B..75 B, widths
// 6..48; player modules: R = B..3 B, widths 152..256).  As library GEMMs (M x N = O x I tiny, K = R ~ 10^6) these took
// 0.9-1.6 ms each - 40 % of a training step - because a 64 x 64 output offers a library kernel only a handful of tiles.
// Here the ROWS are split over the grid: a 256-thread workgroup streams its slice of X and dY through LDS (transposed on
// the way in, so that an MFMA fragment - 8 consecutive k of one column - is one 16 B LDS read), keeps the whole O x (I+1)
// result in MFMA accumulators (v_mfma_f32_16x16x32_bf16; column I of the X tile is all ones, which yields db for free)
// and adds it to dW / db with fp32 atomics at the end.  HBM-bound: R x (I + O) x 2 B read once.
// Wave w owns o-tiles [w*OTW, (w+1)*OTW); IT i-tiles cover I+1 columns.  bf16 inputs, fp32 outputs (zeroed by the caller).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
constexpr int WG_KT = 64;            // rows per stage
constexpr int WG_LD = WG_KT + 8;     // LDS row pitch in elements (144 B: 16 B aligned fragments)

// stage one operand tile: the 64 x W row-major span starting at element `base` (of `total` elements), transposed into
// T[col][k].  NV = number of 16 B vectors per thread.
template <int NV>
__device__ __forceinline__ void wg_load(const unsigned short* __restrict__ src, long base, long total, int W, uint4 (&v)[NV], int tid) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const long f0 = (long)(tid + 256 * j) * 8;
        v[j] = make_uint4(0, 0, 0, 0);
        if (f0 < (long)WG_KT * W) {
            const long g = base + f0;
            if (g + 8 <= total) v[j] = *reinterpret_cast<const uint4*>(src + g);
            else {
                unsigned short e[8];
#pragma unroll
                for (int q = 0; q < 8; q++) e[q] = g + q < total ? src[g + q] : (unsigned short)0;
                v[j] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
            }
        }
    }
}
template <int NV>
__device__ __forceinline__ void wg_store(unsigned short* T, int W, const uint4 (&v)[NV], int tid) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const unsigned f0 = (unsigned)(tid + 256 * j) * 8u;
        if (f0 < (unsigned)(WG_KT * W)) {
            unsigned row = f0 / (unsigned)W, col = f0 - row * (unsigned)W;
            const unsigned w4[4] = { v[j].x, v[j].y, v[j].z, v[j].w };
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (row < (unsigned)WG_KT) T[col * WG_LD + row] = (unsigned short)(w4[q >> 1] >> (16 * (q & 1)));
                if (++col == (unsigned)W) { col = 0; row++; }
            }
        }
    }
}

// ldw / colw: dW is the column window colw .. colw + I - 1 of a weight gradient with leading dimension ldw (0: dW is [O][I]) - a layer
// that multiplies a few conditioning columns with a column slice of a wider weight accumulates straight into the parameter's gradient.
template <int OTW, int IT>
__device__ __forceinline__ void wgrad_body(const unsigned short* __restrict__ X, const unsigned short* __restrict__ dY,
                                           float* __restrict__ dW, float* __restrict__ db, long R, int I, int O, long rows_per_block,
                                           long ldw_, int colw, long bid) {
    constexpr int XC = IT * 16, YC = OTW * 4 * 16;           // padded column counts
    constexpr int NX = (WG_KT * XC / 8 + 255) / 256, NY = (WG_KT * YC / 8 + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned short Xt[XC * WG_LD];
    __shared__ __attribute__((aligned(16))) unsigned short Yt[YC * WG_LD];
    const long ldw = ldw_ ? ldw_ : I;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r_begin = bid * rows_per_block;
    const long r_end = r_begin + rows_per_block < R ? r_begin + rows_per_block : R;
    if (r_begin >= R) return;
    for (int x = tid; x < XC * WG_LD; x += 256) Xt[x] = 0;
    for (int x = tid; x < YC * WG_LD; x += 256) Yt[x] = 0;
    f32x4_t acc[OTW][IT];
#pragma unroll
    for (int a = 0; a < OTW; a++)
#pragma unroll
        for (int b = 0; b < IT; b++) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 vx[NX], vy[NY];
    wg_load<NX>(X, r_begin * I, R * (long)I, I, vx, tid);
    wg_load<NY>(dY, r_begin * O, R * (long)O, O, vy, tid);
    __syncthreads();                                         // zero fill complete
    for (long r0 = r_begin; r0 < r_end; r0 += WG_KT) {
        wg_store<NX>(Xt, I, vx, tid);
        wg_store<NY>(Yt, O, vy, tid);
        if (tid < WG_KT) Xt[I * WG_LD + tid] = (r0 + tid < r_end) ? (unsigned short)0x3F80 : (unsigned short)0;   // bf16 1.0: the bias column
        __syncthreads();
        if (r0 + WG_KT < r_end) {                            // next stage's loads fly during the MFMAs
            wg_load<NX>(X, (r0 + WG_KT) * I, R * (long)I, I, vx, tid);
            wg_load<NY>(dY, (r0 + WG_KT) * O, R * (long)O, O, vy, tid);
        }
        // rows_per_block is a multiple of WG_KT, so only the last block has a partial stage, and its missing rows were
        // loaded as zeros
#pragma unroll
        for (int ks = 0; ks < WG_KT / 32; ks++) {
            const int koff = ks * 32 + 8 * (lane >> 4);
            bf16x8_t bfrag[IT];
#pragma unroll
            for (int b = 0; b < IT; b++) bfrag[b] = *reinterpret_cast<const bf16x8_t*>(&Xt[(b * 16 + (lane & 15)) * WG_LD + koff]);
#pragma unroll
            for (int a = 0; a < OTW; a++) {
                bf16x8_t afrag = *reinterpret_cast<const bf16x8_t*>(&Yt[((wave * OTW + a) * 16 + (lane & 15)) * WG_LD + koff]);
#pragma unroll
                for (int b = 0; b < IT; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[b], acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < OTW; a++)
#pragma unroll
        for (int b = 0; b < IT; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = (wave * OTW + a) * 16 + 4 * (lane >> 4) + r, i = b * 16 + (lane & 15);
                const float v = acc[a][b][r];
                if (o < O && v != 0.0f) {
                    if (i < I) atomicAdd(&dW[(long)o * ldw + colw + i], v);
                    else if (i == I && db != nullptr) atomicAdd(&db[o], v);
                }
            }
}
template <int OTW, int IT>
__global__ __launch_bounds__(256) void k_wgrad(const unsigned short* __restrict__ X, const unsigned short* __restrict__ dY,
                                               float* __restrict__ dW, float* __restrict__ db, long R, int I, int O, long rows_per_block) {
    wgrad_body<OTW, IT>(X, dY, dW, db, R, I, O, rows_per_block, 0L, 0, (long)blockIdx.x);
}


// The same product with the LDS image kept ROW-major and the operand fragments fetched by gfx950's transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group reads a [4 rows][16 columns] block, 8 bytes per lane, and lane i receives column i).
// k_wgrad's transposition on the way IN costs 8 two-byte LDS writes per 16 B of input at a 32-way bank conflict (the lanes
// of a wave write 1 152 B apart): measured 1.3 ms for a 3.9 M x (64 + 192) layer, 1.5 TB/s.  Here the stage is written with
// one 16 B store per 16 B loaded into [16-column tile][k-step of 32 rows][32][16] sub-tiles; within a k-step the fragment's
// eight k are rows 4 g + {0..3} and 16 + 4 g + {0..3} (g = lane >> 4) - a permutation of the contraction index that both
// operands share, and the one for which a wave's read covers 512 contiguous bytes (no bank conflict).
// Requires I % 8 == 0 and O % 8 == 0 (a 16 B vector never straddles a row); the other widths keep k_wgrad.
typedef __attribute__((ext_vector_type(4))) short wg_s4;
constexpr int WG_SUB = 32 * 16 + 16;      // elements per sub-tile (+32 B: neighbouring tiles start 8 banks apart for the stores)
__device__ __forceinline__ int wg_sub_off(int tile, int ks) { return (tile * (WG_KT / 32) + ks) * WG_SUB; }
template <int NV>
__device__ __forceinline__ void wg_store_rows(unsigned short* T, int W, const uint4 (&v)[NV], int tid) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const unsigned f0 = (unsigned)(tid + 256 * j) * 8u;
        if (f0 < (unsigned)(WG_KT * W)) {
            const unsigned row = f0 / (unsigned)W, col = f0 - row * (unsigned)W;          // col is a multiple of 8
            *reinterpret_cast<uint4*>(T + wg_sub_off(col >> 4, row >> 5) + (row & 31) * 16 + (col & 15)) = v[j];
        }
    }
}
__device__ __forceinline__ bf16x8_t wg_frag_tr(const unsigned short* T, int tile, int ks, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const unsigned short* p = T + wg_sub_off(tile, ks) + (4 * g + (i >> 2)) * 16 + (i & 3) * 4;
    union { bf16x8_t f; wg_s4 h[2]; } r;
    r.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)p);
    r.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(p + 16 * 16));
    return r.f;
}

// the 64 x W block of rows r0 .. r0 + 63, columns col0 .. col0 + W - 1 of a row-major matrix with leading dimension ld
// (W, col0, ld multiples of 8: every 16 B vector lies inside one row); rows >= R read as zero
template <int NV>
__device__ __forceinline__ void wg_load_cols(const unsigned short* __restrict__ src, long r0, long R, long ld, int col0, int W, uint4 (&v)[NV], int tid) {
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const unsigned f0 = (unsigned)(tid + 256 * j) * 8u;
        v[j] = make_uint4(0, 0, 0, 0);
        if (f0 < (unsigned)(WG_KT * W)) {
            const unsigned row = f0 / (unsigned)W, col = f0 - row * (unsigned)W;
            if (r0 + row < R) v[j] = *reinterpret_cast<const uint4*>(src + (r0 + row) * ld + col0 + col);
        }
    }
}
template <int OTW, int IT>
__device__ __forceinline__ void wgrad_tr_body(const unsigned short* __restrict__ X, const unsigned short* __restrict__ dY,
                                              float* __restrict__ dW, float* __restrict__ db, long R, int I, int O, long rows_per_block,
                                              long ldx, int col0, long ldw_, int colw, long bid) {
    // ldx != 0: X is a column slice - columns col0 .. col0 + I - 1 of a matrix with leading dimension ldx, and dW the same
    // columns of a weight gradient with that leading dimension (the layers wider than one launch covers are done in slices)
    constexpr int XC = IT * 16, YC = OTW * 4 * 16;           // padded column counts
    constexpr int NX = (WG_KT * XC / 8 + 255) / 256, NY = (WG_KT * YC / 8 + 255) / 256;
    constexpr int XE = IT * (WG_KT / 32) * WG_SUB, YE = OTW * 4 * (WG_KT / 32) * WG_SUB;
    const long ldw = ldw_ ? ldw_ : I;                         // (dW window: see wgrad_body)
    __shared__ __attribute__((aligned(16))) unsigned short Xs[XE];
    __shared__ __attribute__((aligned(16))) unsigned short Ys[YE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r_begin = bid * rows_per_block;
    const long r_end = r_begin + rows_per_block < R ? r_begin + rows_per_block : R;
    if (r_begin >= R) return;
    for (int x = tid; x < XE; x += 256) Xs[x] = 0;
    for (int x = tid; x < YE; x += 256) Ys[x] = 0;
    f32x4_t acc[OTW][IT];
#pragma unroll
    for (int a = 0; a < OTW; a++)
#pragma unroll
        for (int b = 0; b < IT; b++) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 vx[NX], vy[NY];
    if (ldx) wg_load_cols<NX>(X, r_begin, R, ldx, col0, I, vx, tid); else wg_load<NX>(X, r_begin * I, R * (long)I, I, vx, tid);
    wg_load<NY>(dY, r_begin * O, R * (long)O, O, vy, tid);
    __syncthreads();                                         // zero fill complete
    for (long r0 = r_begin; r0 < r_end; r0 += WG_KT) {
        wg_store_rows<NX>(Xs, I, vx, tid);
        wg_store_rows<NY>(Ys, O, vy, tid);
        if (tid < WG_KT) Xs[wg_sub_off(I >> 4, tid >> 5) + (tid & 31) * 16 + (I & 15)] = (r0 + tid < r_end) ? (unsigned short)0x3F80 : (unsigned short)0;   // bf16 1.0: the bias column
        __syncthreads();
        if (r0 + WG_KT < r_end) {                            // next stage's loads fly during the MFMAs
            if (ldx) wg_load_cols<NX>(X, r0 + WG_KT, R, ldx, col0, I, vx, tid); else wg_load<NX>(X, (r0 + WG_KT) * I, R * (long)I, I, vx, tid);
            wg_load<NY>(dY, (r0 + WG_KT) * O, R * (long)O, O, vy, tid);
        }
#pragma unroll
        for (int ks = 0; ks < WG_KT / 32; ks++) {
            bf16x8_t bfrag[IT];
#pragma unroll
            for (int b = 0; b < IT; b++) bfrag[b] = wg_frag_tr(Xs, b, ks, lane);
#pragma unroll
            for (int a = 0; a < OTW; a++) {
                const bf16x8_t afrag = wg_frag_tr(Ys, wave * OTW + a, ks, lane);
#pragma unroll
                for (int b = 0; b < IT; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[b], acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < OTW; a++)
#pragma unroll
        for (int b = 0; b < IT; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = (wave * OTW + a) * 16 + 4 * (lane >> 4) + r, i = b * 16 + (lane & 15);
                const float v = acc[a][b][r];
                if (o < O && v != 0.0f) {
                    if (i < I) atomicAdd(&dW[(long)o * ldw + colw + i], v);
                    else if (i == I && db != nullptr) atomicAdd(&db[o], v);
                }
            }
}

template <int OTW, int IT>
__global__ __launch_bounds__(256) void k_wgrad_tr(const unsigned short* __restrict__ X, const unsigned short* __restrict__ dY,
                                                  float* __restrict__ dW, float* __restrict__ db, long R, int I, int O, long rows_per_block,
                                                  long ldx = 0, int col0 = 0) {
    wgrad_tr_body<OTW, IT>(X, dY, dW, db, R, I, O, rows_per_block, ldx, col0, ldx, col0, (long)blockIdx.x);     // (a column slice of X goes to the same columns of dW)
}
// Several weight gradients of the same tile shape in ONE launch (catan_linear_wgrad_grouped): the action heads' first layers see a
// few thousand to a few ten thousand rows each (the rows whose action type uses the head), and a launch per (head, column slice) -
// 44 of them per minibatch step - is all ramp and tail.  A unit = one (problem, column slice); its blocks follow the previous unit's.
constexpr int WG_MAX_UNITS = 24;
struct WgUnit { const unsigned short* X; const unsigned short* dY; float* dW; float* db; long R; long per; long ldx; long ldw; int I, O, col0, colw, block0, pad_; };
struct WgBatch { WgUnit u[WG_MAX_UNITS]; int n; };
template <int OTW, int IT>
__global__ __launch_bounds__(256) void k_wgrad_tr_grouped(WgBatch b) {
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < b.n; i++) if ((int)blockIdx.x >= b.u[i].block0) k = i;
    const WgUnit& u = b.u[k];
    wgrad_tr_body<OTW, IT>(u.X, u.dY, u.dW, u.db, u.R, u.I, u.O, u.per, u.ldx, u.col0, u.ldw, u.colw, (long)blockIdx.x - u.block0);
}
template <int OTW, int IT>
__global__ __launch_bounds__(256) void k_wgrad_grouped(WgBatch b) {      // widths that are not multiples of 8 (k_wgrad's staging)
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < b.n; i++) if ((int)blockIdx.x >= b.u[i].block0) k = i;
    const WgUnit& u = b.u[k];
    wgrad_body<OTW, IT>(u.X, u.dY, u.dW, u.db, u.R, u.I, u.O, u.per, u.ldw, u.colw, (long)blockIdx.x - u.block0);
}

// ------------------------------------------------------------------------------------------------ tall-skinny linear (forward / dX)
// y[r][n] = sum_k x[r][k] * W[n][k] (+ b[n]) for huge row counts and small widths (K = in <= 128 and a multiple of 8,
// N = out <= 192): the per-tile / per-card layers of the net have 10^6 rows and 16..192 columns, where a library GEMM reaches
// ~40 % of the HBM roofline.  Rows are split over the grid; a wave owns 16-row tiles: its A fragments (8 consecutive k of a
// row) are 16 B global loads straight from x, the whole W sits in B-fragment registers for the kernel's lifetime
// (v_mfma_f32_16x16x32_bf16), and the 16 x N result goes through a small LDS tile so that it leaves as 16 B row stores.
// HBM-bound by construction: R x (K + N) x 2 B moved once.  bf16 in / out, fp32 accumulate; bias optional.
// Optional fused epilogue (mode; N a multiple of 8): 1 = ReLU, 2 = + aux (residual stream), 3 = zero where aux <= 0 (the backward of
// a ReLU whose output is aux) - each applied to the bf16-rounded product exactly as the separate elementwise op would.
__device__ __forceinline__ unsigned short f2bf_nn(float x) { const __hip_bfloat16 h = __float2bfloat16(x); return *reinterpret_cast<const unsigned short*>(&h); }
template <int KS, int NT>
__global__ __launch_bounds__(256) void k_linear_rows(const unsigned short* __restrict__ x, const unsigned short* __restrict__ W,
                                                     const unsigned short* __restrict__ bias, unsigned short* __restrict__ y,
                                                     long R, int K, int N, const unsigned short* __restrict__ aux, int mode) {
    constexpr int NP = NT * 16 + 8;                              // LDS row pitch (elements): 16 B aligned, conflict-light
    __shared__ __attribute__((aligned(16))) unsigned short ot[4][16 * NP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lr = lane & 15, lk = (lane >> 4) * 8;
    // B fragments: lane holds W[n = nt*16 + lr][k = ks*32 + lk .. +7] (zero beyond N / K)
    bf16x8_t bfrag[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            uint4 u = make_uint4(0, 0, 0, 0);
            const int n = nt * 16 + lr, k = ks * 32 + lk;
            if (n < N && k < K) u = *reinterpret_cast<const uint4*>(W + (long)n * K + k);
            bfrag[nt][ks] = *reinterpret_cast<const bf16x8_t*>(&u);
        }
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int n = nt * 16 + lr;
        bv[nt] = (bias != nullptr && n < N) ? __uint_as_float((unsigned)bias[n] << 16) : 0.0f;
    }
    const long tiles = (R + 15) / 16;
    for (long t = (long)blockIdx.x * 4 + wv; t < tiles; t += (long)gridDim.x * 4) {
        const long r0 = t * 16;
        f32x4_t acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            uint4 u = make_uint4(0, 0, 0, 0);
            const int k = ks * 32 + lk;
            if (r0 + lr < R && k < K) u = *reinterpret_cast<const uint4*>(x + (r0 + lr) * (long)K + k);
            const bf16x8_t afrag = *reinterpret_cast<const bf16x8_t*>(&u);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, bfrag[nt][ks], acc[nt], 0, 0, 0);
        }
        // C layout: lane holds rows 4*(lane>>4)+r (r = 0..3), column lane&15 of every n-tile
        unsigned short* o = ot[wv];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const __hip_bfloat16 h = __float2bfloat16(acc[nt][r] + bv[nt]);
                o[(4 * (lane >> 4) + r) * NP + nt * 16 + lr] = *reinterpret_cast<const unsigned short*>(&h);
            }
        __builtin_amdgcn_wave_barrier();
        // row stores: 16 rows x N elements; 8-element (16 B) chunks where the row pitch allows, else element-wise
        if ((N & 7) == 0) {
            const int cpr = N / 8;                               // chunks per row
            for (int c = lane; c < 16 * cpr; c += 64) {
                const int row = c / cpr, ch = c - row * cpr;
                if (r0 + row >= R) continue;
                uint4 v = *reinterpret_cast<const uint4*>(o + row * NP + ch * 8);
                if (mode != 0) {                                 // fused epilogue on the (already bf16-rounded) result, as the separate op would see it
                    uint4 a = make_uint4(0, 0, 0, 0);
                    if (mode >= 2) a = *reinterpret_cast<const uint4*>(aux + (r0 + row) * (long)N + ch * 8);
                    unsigned* vw = reinterpret_cast<unsigned*>(&v); const unsigned* aw = reinterpret_cast<const unsigned*>(&a);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float lo = __uint_as_float(vw[i] << 16), hi = __uint_as_float(vw[i] & 0xFFFF0000u);
                        const float alo = __uint_as_float(aw[i] << 16), ahi = __uint_as_float(aw[i] & 0xFFFF0000u);
                        if (mode == 1) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }                       // ReLU
                        else if (mode == 2) { lo += alo; hi += ahi; }                                        // + residual
                        else { lo = alo > 0.f ? lo : 0.f; hi = ahi > 0.f ? hi : 0.f; }                      // ReLU backward: aux = the ReLU's output
                        vw[i] = pk_bf(lo, hi);
                    }
                }
                *reinterpret_cast<uint4*>(y + (r0 + row) * (long)N + ch * 8) = v;
            }
        } else {
            for (int c = lane; c < 16 * N; c += 64) {
                const int row = c / N, col = c - row * N;
                if (r0 + row < R) y[(r0 + row) * (long)N + col] = o[row * NP + col];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LSTM cell (the optional `include_lstm` path: RL/models/policy.py:36-45,113-166 uses torch.nn.LSTM, gate order i, f, g, o).
// The two GEMMs of a step (x W_ih^T over all steps at once, h W_hh^T per step) stay library GEMMs; everything between them
// is this one pass: gates = gx + gh, c = sigmoid(f) * (c_prev * mask) + sigmoid(i) * tanh(g), h = sigmoid(o) * tanh(c).
// A thread owns 4 consecutive hidden units of one row: four 8 B (bf16) / 16 B (fp32) loads per gate tensor, all coalesced.
// HBM-bound: (2 * 4L * sizeof(T) + 3 * 4L) bytes per row forward.  The backward recomputes the activations from the same
// inputs and writes the pre-activation gate gradient (it is the gradient of gx and of gh alike) and dc_prev.
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f / (1.0f + __expf(2.0f * x)); }

template <class T>
__global__ __launch_bounds__(256) void k_lstm_cell_fwd(const T* __restrict__ gx, const T* __restrict__ gh, const float* __restrict__ c_prev,
                                                       const float* __restrict__ mask, float* __restrict__ h_out,
                                                       float* __restrict__ c_out, long n, int L) {
    const int q = L / 4;
    const long items = n * q;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const long row = it / q;
        const int j = (int)(it - row * q) * 4;
        const T* px = gx + row * 4 * L + j;
        const T* ph = gh + row * 4 * L + j;
        float a[4][4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            RowVec<T, 4>::load(px + k * L, a[k]);
            RowVec<T, 4>::load(ph + k * L, b);
#pragma unroll
            for (int e = 0; e < 4; e++) a[k][e] += b[e];
        }
        float cp[4], h[4], c[4];
        RowVec<float, 4>::load(c_prev + row * L + j, cp);
        const float mm = mask ? mask[row] : 1.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            c[e] = sigmoid_f(a[1][e]) * (cp[e] * mm) + sigmoid_f(a[0][e]) * tanh_f(a[2][e]);
            h[e] = sigmoid_f(a[3][e]) * tanh_f(c[e]);
        }
        RowVec<float, 4>::store(c_out + row * L + j, c);
        RowVec<float, 4>::store(h_out + row * L + j, h);
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_lstm_cell_bwd(const T* __restrict__ gx, const T* __restrict__ gh, const float* __restrict__ c_prev,
                                                       const float* __restrict__ mask, const float* __restrict__ dh,
                                                       const float* __restrict__ dc_out, T* __restrict__ dgates,
                                                       float* __restrict__ dc_prev, long n, int L) {
    const int q = L / 4;
    const long items = n * q;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        const long row = it / q;
        const int j = (int)(it - row * q) * 4;
        const T* px = gx + row * 4 * L + j;
        const T* ph = gh + row * 4 * L + j;
        float a[4][4], b[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            RowVec<T, 4>::load(px + k * L, a[k]);
            RowVec<T, 4>::load(ph + k * L, b);
#pragma unroll
            for (int e = 0; e < 4; e++) a[k][e] += b[e];
        }
        float cp[4], gh_[4], gc[4], dcp[4];
        RowVec<float, 4>::load(c_prev + row * L + j, cp);
        RowVec<float, 4>::load(dh + row * L + j, gh_);
        RowVec<float, 4>::load(dc_out + row * L + j, gc);
        const float mm = mask ? mask[row] : 1.0f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float i = sigmoid_f(a[0][e]), f = sigmoid_f(a[1][e]), g = tanh_f(a[2][e]), o = sigmoid_f(a[3][e]);
            const float cin = cp[e] * mm;
            const float tc = tanh_f(f * cin + i * g);
            const float dc = gc[e] + gh_[e] * o * (1.0f - tc * tc);
            a[0][e] = dc * g * i * (1.0f - i);
            a[1][e] = dc * cin * f * (1.0f - f);
            a[2][e] = dc * i * (1.0f - g * g);
            a[3][e] = gh_[e] * tc * o * (1.0f - o);
            dcp[e] = dc * f * mm;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) RowVec<T, 4>::store(dgates + row * 4 * L + j + k * L, a[k]);
        RowVec<float, 4>::store(dc_prev + row * L + j, dcp);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Masked categorical head (RL/distributions.py:10-40 + the sampling / log-prob / entropy calls of
// RL/models/action_heads_module.py:202-256): logp = log_softmax(logits + log(mask)); action = given | arg-max | inverse-CDF
// sample with the caller's uniform u; outputs the action, its log-prob and the row entropy (-sum p logp over p > 0).
// As torch ops this was a dozen launches per head, twenty head evaluations per policy call - a quarter of the launches of a
// call that is launch-bound at rollout width.  One lane per row, three passes over the K <= 73 logits of the row (the rows
// of neighbouring lanes are adjacent, so every cache line is used completely over the k loop).
__global__ __launch_bounds__(256) void k_categorical_fwd(const float* __restrict__ logits, const float* __restrict__ mask, long mask_ld,
                                                         const long long* __restrict__ given, const float* __restrict__ u,
                                                         long long* __restrict__ action, float* __restrict__ logp,
                                                         float* __restrict__ entropy, float* __restrict__ lse_out, long B, int K) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= B) return;
    const float* z = logits + row * K;
    const float* m = mask + row * mask_ld;
    float mx = -INFINITY;
    int amax = 0;
    for (int k = 0; k < K; k++) if (m[k] > 0.0f && z[k] > mx) { mx = z[k]; amax = k; }
    float sum = 0.0f;
    for (int k = 0; k < K; k++) if (m[k] > 0.0f) sum += __expf(z[k] - mx);
    const float lse = mx + __logf(sum);
    const float thr = u ? u[row] : 0.0f;
    float cdf = 0.0f, ent = 0.0f;
    int pick = -1, last = amax;
    for (int k = 0; k < K; k++) {
        if (!(m[k] > 0.0f)) continue;
        const float lp = z[k] - lse, p = __expf(lp);
        if (p > 0.0f) ent -= p * lp;
        cdf += p;
        last = k;
        if (pick < 0 && cdf > thr) pick = k;
    }
    int a = given ? (int)given[row] : (u ? (pick >= 0 ? pick : last) : amax);
    a = min(max(a, 0), K - 1);
    action[row] = a;
    logp[row] = (m[a] > 0.0f ? z[a] : -INFINITY) - lse;
    entropy[row] = ent;
    lse_out[row] = lse;
}
// d logits = dlogp * (onehot(a) - p) - dent * p * (logp + H)
__global__ __launch_bounds__(256) void k_categorical_bwd(const float* __restrict__ logits, const float* __restrict__ mask, long mask_ld,
                                                         const long long* __restrict__ action, const float* __restrict__ lse,
                                                         const float* __restrict__ entropy, const float* __restrict__ dlogp,
                                                         const float* __restrict__ dent, float* __restrict__ dlogits, long B, int K) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= B) return;
    const float* z = logits + row * K;
    const float* m = mask + row * mask_ld;
    float* dz = dlogits + row * K;
    const int a = (int)action[row];
    const float l = lse[row], H = entropy[row], gl = dlogp[row], ge = dent[row];
    for (int k = 0; k < K; k++) {
        float g = 0.0f;
        if (m[k] > 0.0f) {
            const float lp = z[k] - l, p = __expf(lp);
            g = gl * ((k == a ? 1.0f : 0.0f) - p);
            if (p > 0.0f) g -= ge * p * (lp + H);
        }
        dz[k] = g;
    }
}


// The same head with the mask read as BITS of the env's packed mask rows (uint32 [n][pitch], bit i of the flat 325-entry mask = word
// i >> 5, bit i & 31) - the learner's evaluation of given actions.  As float windows of an expanded [rows, 325] matrix every lane's K mask
// values sat 1 300 bytes from its neighbour's (a cache line per lane and pass), on top of expanding 170 000 x 325 floats per minibatch step
// and gathering them per head.  Row j of the launch reads packed row rows[j] (rows == nullptr: j).  Up to three SEGMENTS of rows use different
// bit offsets (rows j < n0: segment 0, j < n1: segment 1, else 2) - the heads whose mask row depends on the action type (corner: settlement /
// city; relative player: propose / steal; resource: exchange / year of plenty / monopoly) run on rows sorted by type - and a segment may AND
// a second bit range in (aoff >= 0: the env's "card playable" row x the card's resource row, build_agent_model.py:113-124).
// given: int64, given_ld elements between rows (a column of the gathered action rows).
struct CatBits { const unsigned* pm; long pitch; const long long* rows; int n0, n1; int off[3], aoff[3]; };
__device__ __forceinline__ bool cat_bit(const unsigned* w, int off, int aoff, int k) {
    bool b = (w[(off + k) >> 5] >> ((off + k) & 31)) & 1u;
    if (aoff >= 0) b = b && ((w[(aoff + k) >> 5] >> ((aoff + k) & 31)) & 1u);
    return b;
}
__global__ __launch_bounds__(256) void k_categorical_bits_fwd(const float* __restrict__ logits, CatBits mb, const long long* __restrict__ given, long given_ld,
                                                              long long* __restrict__ action, float* __restrict__ logp, float* __restrict__ entropy,
                                                              float* __restrict__ lse_out, long B, int K) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= B) return;
    const float* z = logits + row * K;
    const unsigned* w = mb.pm + (mb.rows ? mb.rows[row] : row) * mb.pitch;
    const int seg = row < mb.n0 ? 0 : (row < mb.n1 ? 1 : 2);
    const int off = mb.off[seg], aoff = mb.aoff[seg];
    float mx = -INFINITY;
    int amax = 0;
    for (int k = 0; k < K; k++) if (cat_bit(w, off, aoff, k) && z[k] > mx) { mx = z[k]; amax = k; }
    float sum = 0.0f;
    for (int k = 0; k < K; k++) if (cat_bit(w, off, aoff, k)) sum += __expf(z[k] - mx);
    const float lse = mx + __logf(sum);
    float ent = 0.0f;
    for (int k = 0; k < K; k++) {
        if (!cat_bit(w, off, aoff, k)) continue;
        const float lp = z[k] - lse, p = __expf(lp);
        if (p > 0.0f) ent -= p * lp;
    }
    int a = given ? (int)given[row * given_ld] : amax;
    a = min(max(a, 0), K - 1);
    action[row] = a;
    logp[row] = (cat_bit(w, off, aoff, a) ? z[a] : -INFINITY) - lse;
    entropy[row] = ent;
    lse_out[row] = lse;
}
__global__ __launch_bounds__(256) void k_categorical_bits_bwd(const float* __restrict__ logits, CatBits mb, const long long* __restrict__ action,
                                                              const float* __restrict__ lse, const float* __restrict__ entropy, const float* __restrict__ dlogp,
                                                              const float* __restrict__ dent, float* __restrict__ dlogits, long B, int K) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= B) return;
    const float* z = logits + row * K;
    const unsigned* w = mb.pm + (mb.rows ? mb.rows[row] : row) * mb.pitch;
    const int seg = row < mb.n0 ? 0 : (row < mb.n1 ? 1 : 2);
    const int off = mb.off[seg], aoff = mb.aoff[seg];
    float* dz = dlogits + row * K;
    const int a = (int)action[row];
    const float l = lse[row], H = entropy[row], gl = dlogp[row], ge = dent[row];
    for (int k = 0; k < K; k++) {
        float g = 0.0f;
        if (cat_bit(w, off, aoff, k)) {
            const float lp = z[k] - l, p = __expf(lp);
            g = gl * ((k == a ? 1.0f : 0.0f) - p);
            if (p > 0.0f) g -= ge * p * (lp + H);
        }
        dz[k] = g;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention on MFMA for bf16 (L = 19 tiles x 4 heads x 16; L = 25 cards x 4 heads x 4 with a key-length mask): the VALU
// kernels above spend 115 k FMAs per tile sequence in the backward and ran 4-5x off the HBM floor (config-3 minibatch: 19 of
// 86 ms in attention).  One wave = one sequence,
// one v_mfma_f32_32x32x16_bf16 per product, everything TRANSPOSED so that no operand ever changes lanes:
//   S^T = K Q^T           A = K rows j, B = Q rows i: 16-byte reads of the lane's row of the LDS image (8 of the 16 head dims per half)
//   C layout of a 32x32 product: lane holds column (lane & 31) and rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5), r = 0..15
//   -> a lane owns query i and sixteen of the 32 (padded) keys: the softmax over keys is in-lane plus ONE xor-32 exchange
//   O^T = V^T P^T         B = P^T straight from those registers: the contraction index may be permuted freely as long as A
//                         uses the same permutation - k-step s, half h, element t  <->  key 16 s + 4 h + t (t < 4),
//                         16 s + 8 + 4 h + (t - 4) (t >= 4), i.e. registers 8 s .. 8 s + 7; A = V^T rows d through the transposing
//                         LDS read of the token-major V columns (ds_read_b64_tr_b16: four keys' dim d per instruction)
//   O^T lands as lane = query i, rows = head dims: two 8-byte LDS stores per head into the image; whole rows leave.
// The backward adds the same products in the other orientation (lane = key j) for dK / dV, exchanging only the per-query
// softmax statistics through LDS.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ unsigned short f2bf(float x) { const __hip_bfloat16 h = __float2bfloat16(x); return *reinterpret_cast<const unsigned short*>(&h); }
__device__ __forceinline__ bf16x8_t pack_bf8(const float* v) {
    union { bf16x8_t f; unsigned u[4]; } r;
#pragma unroll
    for (int t = 0; t < 4; t++) r.u[t] = pk_bf(v[2 * t], v[2 * t + 1]);
    return r.f;
}
__device__ __forceinline__ bf16x8_t zero_bf8() { union { bf16x8_t f; uint4 u; } r; r.u = make_uint4(0, 0, 0, 0); return r.f; }
// the 8 k-elements of this lane half for one head slice of a row: head dims 8 hf .. 8 hf + 7 (HD = 16) or dims 0..3 + zeros (HD = 4)
template <int HD>
__device__ __forceinline__ bf16x8_t ld_frag(const unsigned short* head, int hf, bool ok) {
    union { bf16x8_t f; uint4 u; uint2 v[2]; } r;
    r.u = make_uint4(0, 0, 0, 0);
    if constexpr (HD == 16) { if (ok) r.u = *reinterpret_cast<const uint4*>(head + 8 * hf); }
    else { if (ok && hf == 0) r.v[0] = *reinterpret_cast<const uint2*>(head); }
    return r.f;
}
// rows d < HD of a [d][col] product sit in registers 0..7 (HD = 16: d = 4 hf + r, 8 + 4 hf + (r - 4)) or 0..3 of half 0 (HD = 4)
template <int HD>
__device__ __forceinline__ void st_head(unsigned short* head, const f32x16_t& c, int hf) {
    if constexpr (HD == 16) {
        *reinterpret_cast<uint2*>(head + 4 * hf) = make_uint2(pk_bf(c[0], c[1]), pk_bf(c[2], c[3]));
        *reinterpret_cast<uint2*>(head + 8 + 4 * hf) = make_uint2(pk_bf(c[4], c[5]), pk_bf(c[6], c[7]));
    } else {
        if (hf == 0) *reinterpret_cast<uint2*>(head) = make_uint2(pk_bf(c[0], c[1]), pk_bf(c[2], c[3]));
    }
}

// The backward works from row-major LDS images of Q, K, V, dO (coalesced 16-byte row pieces in, 16-byte stores to LDS) and collects
// dQ | dK | dV in an LDS image that leaves as whole rows: with every lane fetching ITS row's 16-byte slice per head straight from
// HBM and storing 8-byte slices back, a sequence cost 49 memory instructions that touch 32 cache lines each.
template <int D> constexpr int at_rp() { return D + 8; }             // row pitch of an image (elements): 16-byte aligned rows
template <int L, int D>
__device__ __forceinline__ void stage_rows(unsigned short* dst, const unsigned short* src, int row_stride, int lane) {
    constexpr int CH = D / 8;
    for (int x = lane; x < L * CH; x += 64) {
        const int j = x / CH, c = x - j * CH;
        *reinterpret_cast<uint4*>(dst + j * at_rp<D>() + c * 8) = *reinterpret_cast<const uint4*>(src + j * row_stride + c * 8);
    }
}
// A operand of a product that contracts over the image's ROWS: column `col` (this lane's dim), the 8 rows of k-step s for this lane
// half in the order the probabilities' registers have them (16 s + 4 hf + {0..3}, 16 s + 8 + 4 hf + {0..3}); rows >= L are zeros
template <int L, int D>
__device__ __forceinline__ bf16x8_t ld_gather(const unsigned short* col, int s, int hf) {
    union { bf16x8_t f; unsigned short u[8]; } r;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int row = 16 * s + 8 * (e >> 2) + 4 * hf + (e & 3);
        r.u[e] = row < L ? col[row * at_rp<D>()] : (unsigned short)0;
    }
    return r.f;
}
// ... the same fragment through gfx950's transposing LDS read (16 dims per head): a 16-lane group hands ds_read_tr16_b64 the
// addresses of rows base .. base + 3 (lane i: row base + i / 4, dims 4 (i % 4) ..) and lane i gets dim i of those four rows - one
// instruction per 4-row run instead of four 2-byte reads.  The images have 32 rows (rows >= L are zero).
// hi: the row offset of the second run (8; 0 where those rows lie beyond the image - their coefficients are exact zeros, any
// finite rows do).
__device__ __forceinline__ bf16x8_t ld_gather_tr(const unsigned short* img_head, int rp, int s, int lane, int hi = 8) {
    const int i = lane & 15, hf = lane >> 5;
    const unsigned short* p = img_head + (16 * s + 4 * hf + (i >> 2)) * rp + (i & 3) * 4;
    union { bf16x8_t f; wg_s4 h[2]; } r;
    r.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)p);
    r.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(p + hi * rp));
    return r.f;
}
// The forward: the sequence's whole [L][3][D] block (contiguous in qkv) is staged row-major into an LDS image with coalesced 16-byte
// loads; the K / Q fragments are 16-byte LDS reads of the lane's row, the V^T fragments come through the transposing LDS read (HD = 16)
// or 2-byte gathers (HD = 4); head h's output replaces the (dead) Q columns of head h in the image, and the L x D result leaves as
// whole rows.  (Every lane fetching its row's 16-byte slices per head from HBM and storing 8-byte slices back was 8 + 8 memory
// instructions per sequence that each touched up to 38 cache lines: 0.74 ms per 204 800 sequences.)
template <int L, int H, int HD>
__global__ __launch_bounds__(256) void k_attn_mfma_fwd(const unsigned short* __restrict__ qkv, const int* __restrict__ lens,
                                                       unsigned short* __restrict__ out, long B) {
    constexpr int D = H * HD, NR = L <= 24 ? 12 : 16;
    static_assert(L <= 32 && D % 8 == 0 && (HD == 16 || HD == 4), "one 32x32 tile per product");
    constexpr bool TR = HD == 16;                            // transposing LDS reads (whole 16-dim heads)
    constexpr int RP = 3 * D + 8;                            // row pitch of the image (elements): 16-byte aligned rows
    constexpr int IR = TR ? (L <= 24 ? 24 : 32) : L;         // rows of the image (TR: zero rows behind the sequence, see k_attn_mfma_bwd)
    constexpr int HI1 = IR == 24 ? 0 : 8;
    __shared__ __attribute__((aligned(16))) unsigned short Img[4][IR * RP];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, hf = lane >> 5, c31 = lane & 31;
    const long b = (long)blockIdx.x * 4 + wv;
    if (b >= B) return;
    unsigned short* img = Img[wv];
    if (TR) for (int x = lane; x < (IR - L) * RP / 8; x += 64) reinterpret_cast<uint4*>(img + L * RP)[x] = make_uint4(0, 0, 0, 0);
    const unsigned short* base = qkv + b * (long)(L * 3 * D);
    constexpr int CR = 3 * D / 8;                            // 16-byte pieces per row
    for (int x = lane; x < L * CR; x += 64) {
        const int j = x / CR, c = x - j * CR;
        *reinterpret_cast<uint4*>(img + j * RP + c * 8) = *reinterpret_cast<const uint4*>(base + (long)x * 8);
    }
    __builtin_amdgcn_wave_barrier();
    const bool rowok = c31 < L;
    const int len = lens ? lens[b] : L;
    const float C = (HD == 16 ? 0.25f : 0.5f) * 1.44269504088896340736f;          // 1 / sqrt(HD), in the exponent of 2^x
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int rrow = (rowok ? c31 : 0) * RP;
#pragma unroll
    for (int h = 0; h < H; h++) {
        const bf16x8_t ka = ld_frag<HD>(img + rrow + D + h * HD, hf, rowok);
        const bf16x8_t qb = ld_frag<HD>(img + rrow + h * HD, hf, rowok);
        bf16x8_t va[2];
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if constexpr (TR) va[s] = ld_gather_tr(img + 2 * D + h * HD, RP, s, lane, s == 1 ? HI1 : 8);
            else {
                union { bf16x8_t f; unsigned short u[8]; } r;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int row = 16 * s + 8 * (e >> 2) + 4 * hf + (e & 3);
                    r.u[e] = row < L ? img[row * RP + 2 * D + h * HD + (c31 % HD)] : (unsigned short)0;
                }
                va[s] = r.f;
            }
        }
        const f32x16_t st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qb, zero16, 0, 0, 0);   // S^T[j][i]
        // registers 12..15 hold keys 24..31: beyond every key when L <= 24 (the 19-token tile sequences) - not computed
        float p[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = (r & 3) + 8 * (r >> 2) + 4 * hf;
            p[r] = j < len ? st[r] : -INFINITY;
            mx = fmaxf(mx, p[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mc = -mx * C;
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < NR; r++) { p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[r], C, mc)); sum += p[r]; }
        sum += __shfl_xor(sum, 32);
        const float inv = __builtin_amdgcn_rcpf(sum);
        f32x16_t ot = zero16;
#pragma unroll
        for (int s = 0; s < 2; s++)
            ot = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c31 < HD ? va[s] : zero_bf8(), pack_bf8(p + 8 * s), ot, 0, 0, 0);   // O^T[d][i]
#pragma unroll
        for (int r = 0; r < 8; r++) ot[r] *= inv;
        if (rowok) st_head<HD>(img + rrow + h * HD, ot, hf);           // over head h's Q columns: every lane holds its fragment of them
    }
    __builtin_amdgcn_wave_barrier();
    {
        constexpr int CH = D / 8;
        unsigned short* ob = out + b * (long)(L * D);
        for (int x = lane; x < L * CH; x += 64) {
            const int j = x / CH, c = x - j * CH;
            *reinterpret_cast<uint4*>(ob + j * D + c * 8) = *reinterpret_cast<const uint4*>(img + j * RP + c * 8);
        }
    }
}

// dqkv [B][L][3][D] from dout [B][L][D]; the probabilities are recomputed in both orientations (see above)
// LENS = false: no key mask (lens == nullptr): the length is the constant L - the masks of the keys behind the sequence fold away, and the
// key lanes >= L need no zeroing at all (their dK / dV columns are never stored).  The 1 / sqrt(HD) of dS is a power of two: it is applied to
// the 8 output registers of dQ / dK instead of the 12 dS values, bit-identically.
template <int L, int H, int HD, bool LENS>
__global__ __launch_bounds__(256) void k_attn_mfma_bwd(const unsigned short* __restrict__ qkv, const int* __restrict__ lens,
                                                       const unsigned short* __restrict__ dout, unsigned short* __restrict__ dqkv, long B) {
    constexpr int D = H * HD, NR = L <= 24 ? 12 : 16;      // registers 12..15 = keys / queries 24..31: beyond the sequence when L <= 24
    constexpr int RP = at_rp<D>();
    constexpr bool TR = HD == 16;                            // transposing LDS reads (whole 16-dim heads)
    // rows of an image (TR: padded with zero rows).  L <= 24: 24 rows - the products' rows 24..31 meet the probabilities' registers
    // 12..15, exact zeros, so their fragments re-read rows 16..23 (HI1 = 0): 43 instead of 57 KB of LDS = three workgroups per CU
    constexpr int IR = TR ? (L <= 24 ? 24 : 32) : L;
    constexpr int HI1 = IR == 24 ? 0 : 8;                    // second 4-row run of k-step 1
    __shared__ __attribute__((aligned(16))) unsigned short Tt[4][3][IR * RP];            // K, Q, dO row-major: [key / query][dim]
    __shared__ __attribute__((aligned(16))) float Stat[4][3][32];                        // per query: -max scale log2(e), 1 / row sum, delta
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, hf = lane >> 5, c31 = lane & 31;
    const long b = (long)blockIdx.x * 4 + wv;
    if (b >= B) return;
    unsigned short* kt = Tt[wv][0]; unsigned short* qt = Tt[wv][1]; unsigned short* dot = Tt[wv][2];
    const unsigned short* base = qkv + b * (long)(L * 3 * D);
    const unsigned short* dob = dout + b * (long)(L * D);
    unsigned short* gb = dqkv + b * (long)(L * 3 * D);
    if (TR) {                                                // the zero rows L .. 31 of K, Q, dO (V is read by rows only)
        for (int x = lane; x < 3 * (IR - L) * RP / 8; x += 64) {
            const int t = x / ((IR - L) * RP / 8), o = x - t * ((IR - L) * RP / 8);
            reinterpret_cast<uint4*>(Tt[wv][t] + L * RP)[o] = make_uint4(0, 0, 0, 0);
        }
    }
    stage_rows<L, D>(kt, base + D, 3 * D, lane);
    stage_rows<L, D>(qt, base, 3 * D, lane);
    stage_rows<L, D>(dot, dob, D, lane);
    __builtin_amdgcn_wave_barrier();
    const bool rowok = c31 < L;
    const int len = LENS ? lens[b] : L;
    const float scale = HD == 16 ? 0.25f : 0.5f;
    float* stat = &Stat[wv][0][0];
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int rrow = (rowok ? c31 : 0) * RP;                 // this lane's row of the images (rows >= L: zeros below)
    bf16x8_t qr_[H], kr_[H], vr_[H], gr_[H];
#pragma unroll
    for (int h = 0; h < H; h++) {
        qr_[h] = ld_frag<HD>(qt + rrow + h * HD, hf, rowok);                                         // row c31 of Q, K, V, dO
        kr_[h] = ld_frag<HD>(kt + rrow + h * HD, hf, rowok);
        vr_[h] = ld_frag<HD>(base + (c31 * 3 + 2) * D + h * HD, hf, rowok);
        gr_[h] = ld_frag<HD>(dot + rrow + h * HD, hf, rowok);
    }
#pragma unroll
    for (int h = 0; h < H; h++) {
        const bf16x8_t qr = qr_[h], kr = kr_[h], vr = vr_[h], gr = gr_[h];
        const int tcol = h * HD + (c31 % HD);                                                      // this lane's dim d: a column of the images
        f32x16_t dq_keep;
        // ---- lane = query i, registers = keys j
        {
            const f32x16_t st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kr, qr, zero16, 0, 0, 0);    // S^T[j][i]
            const f32x16_t dpt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vr, gr, zero16, 0, 0, 0);   // dP^T[j][i] = sum_d V[j][d] dO[i][d]
            float p[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const int j = (r & 3) + 8 * (r >> 2) + 4 * hf;
                p[r] = j < len ? st[r] : -INFINITY;
                mx = fmaxf(mx, p[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // exp(scale (s - max)) = 2^(s C - max C): one fma and v_exp_f32 per probability; 1 / sum by v_rcp_f32
            const float C = scale * 1.44269504088896340736f, mc = -mx * C;
            float sum = 0.0f;
#pragma unroll
            for (int r = 0; r < NR; r++) { p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[r], C, mc)); sum += p[r]; }
            sum += __shfl_xor(sum, 32);
            const float inv = __builtin_amdgcn_rcpf(sum);
            float delta = 0.0f;
#pragma unroll
            for (int r = 0; r < NR; r++) { p[r] *= inv; delta += p[r] * dpt[r]; }
            delta += __shfl_xor(delta, 32);
            if (hf == 0) { stat[c31] = mc; stat[32 + c31] = inv; stat[64 + c31] = delta; }
#pragma unroll
            for (int r = 0; r < NR; r++) p[r] = p[r] * (dpt[r] - delta);                             // dS^T[j][i] / scale
            f32x16_t dq = zero16;
#pragma unroll
            for (int s = 0; s < 2; s++)
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c31 < HD ? (TR ? ld_gather_tr(kt + h * HD, RP, s, lane, s == 1 ? HI1 : 8) : ld_gather<L, D>(kt + tcol, s, hf)) : zero_bf8(), pack_bf8(p + 8 * s), dq, 0, 0, 0);   // dQ^T[d][i]
#pragma unroll
            for (int r = 0; r < 8; r++) dq[r] *= scale;
            dq_keep = dq;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lane = key j, registers = queries i
        {
            const f32x16_t s2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qr, kr, zero16, 0, 0, 0);     // S[i][j]
            const f32x16_t dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gr, vr, zero16, 0, 0, 0);     // dP[i][j] = sum_d dO[i][d] V[j][d]
            const bool keyok = !LENS || c31 < len;              // a masked key has probability 0 for every query
            const float C2 = scale * 1.44269504088896340736f;
            float p[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ds[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q4 = 0; q4 < NR / 4; q4++) {                    // registers 4 q4 .. 4 q4 + 3 <-> queries 8 q4 + 4 hf .. + 3
                const float4 m4 = *reinterpret_cast<const float4*>(stat + 8 * q4 + 4 * hf);
                const float4 i4 = *reinterpret_cast<const float4*>(stat + 32 + 8 * q4 + 4 * hf);
                const float4 d4 = *reinterpret_cast<const float4*>(stat + 64 + 8 * q4 + 4 * hf);
                const float mm[4] = { m4.x, m4.y, m4.z, m4.w }, ii[4] = { i4.x, i4.y, i4.z, i4.w }, dd[4] = { d4.x, d4.y, d4.z, d4.w };
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int r = 4 * q4 + e;
                    p[r] = keyok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s2[r], C2, mm[e])) * ii[e] : 0.0f;   // mm = -max C (the query's)
                    ds[r] = p[r] * (dp[r] - dd[e]);
                }
            }
            f32x16_t dk = zero16, dv = zero16;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c31 < HD ? (TR ? ld_gather_tr(qt + h * HD, RP, s, lane, s == 1 ? HI1 : 8) : ld_gather<L, D>(qt + tcol, s, hf)) : zero_bf8(), pack_bf8(ds + 8 * s), dk, 0, 0, 0);   // dK^T[d][j]
                dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c31 < HD ? (TR ? ld_gather_tr(dot + h * HD, RP, s, lane, s == 1 ? HI1 : 8) : ld_gather<L, D>(dot + tcol, s, hf)) : zero_bf8(), pack_bf8(p + 8 * s), dv, 0, 0, 0);   // dV^T[d][j]
            }
#pragma unroll
            for (int r = 0; r < 8; r++) dk[r] *= scale;
            // head h's columns of the Q, K and dO images are dead now (this lane's fragments of them are in registers): dQ, dK, dV of the
            // head take their place, and the three images leave as whole rows below - 8-byte slices stored straight to dqkv were 24 store
            // instructions per sequence that each touched up to 38 cache lines
            if (rowok) {
                st_head<HD>(qt + rrow + h * HD, dq_keep, hf);
                st_head<HD>(kt + rrow + h * HD, dk, hf);
                st_head<HD>(dot + rrow + h * HD, dv, hf);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    {
        constexpr int CH = D / 8;                                                // 16-byte pieces per image row
        for (int x = lane; x < L * 3 * CH; x += 64) {
            const int j = x / (3 * CH), c = x - j * (3 * CH), img = c / CH, cc = c - img * CH;
            const unsigned short* src = (img == 0 ? qt : img == 1 ? kt : dot) + j * RP + cc * 8;
            *reinterpret_cast<uint4*>(gb + j * 3 * D + c * 8) = *reinterpret_cast<const uint4*>(src);
        }
    }
}


// ------------------------------------------------------------------------------------------------ card-list summary
// The dev-card list modules of RL/models/player_modules.py:55-69 (embedding -> 4-head attention with key mask -> out
// projection -> LayerNorm -> zero the padding -> sum over the list), one lane per list.  A list holds at most 6 distinct
// card ids, and everything computed per token depends only on the token's id and on how many valid tokens of each id the
// list has: the softmax over the keys j of s(id_i, id_j) equals the softmax over the ids b of s(id_i, b) + log(count_b).
// So the kernel reads the 25 ids, counts them, and evaluates <= 6 query classes against <= 6 key classes from tiny tables:
//   S[h][a][b] = q_h(a) . k_h(b) / sqrt(hd)   (4 x 6 x 6)      V[b][16]      out-projection W[16][16], bias     LayerNorm w, b
// (the tables are functions of the embedding and the Q/K/V weights - 6 x 16 numbers through 16 x 48 - and are built by
// torch on the host side of the call, so their gradients flow on from dS / dV by autograd).  Traffic: 26 B in, 64 B out per
// list instead of a dozen [B, 25, 16..48] tensors.  ids: element size `esz` (1, 4 or 8 bytes), row pitch `pitch` elements.
constexpr int CS_V = 6, CS_H = 4, CS_HD = 4, CS_D = 16, CS_L = 25;
struct CardTables { float S[CS_H][CS_V][CS_V]; float Vt[CS_V][CS_D]; float W[CS_D][CS_D]; float bo[CS_D]; float lw[CS_D]; float lb[CS_D]; };
constexpr int CS_NPAR = CS_H * CS_V * CS_V + CS_V * CS_D + CS_D * CS_D + 3 * CS_D;     // 544 floats, in this order
static_assert(sizeof(CardTables) == CS_NPAR * sizeof(float), "CardTables is the flat parameter block");

DEVI void cs_counts(const void* __restrict__ ids, int esz, long pitch, long r, int len, float (&cnt)[CS_V]) {
#pragma unroll
    for (int a = 0; a < CS_V; a++) cnt[a] = 0.f;
    for (int i = 0; i < CS_L && i < len; i++) {
        long id;
        if (esz == 1) id = reinterpret_cast<const signed char*>(ids)[r * pitch + i];
        else if (esz == 4) id = reinterpret_cast<const int*>(ids)[r * pitch + i];
        else id = reinterpret_cast<const long long*>(ids)[r * pitch + i];
#pragma unroll
        for (int a = 0; a < CS_V; a++) cnt[a] += (id == a) ? 1.f : 0.f;
    }
}
// one query class: attention output, projection, LayerNorm.  p[h][b] (softmax), attn[16], xhat[16], rstd are kept for the backward
DEVI void cs_class_fwd(const CardTables& T, const float (&logc)[CS_V], int a, float eps, float (&p)[CS_H][CS_V], float (&attn)[CS_D],
                       float (&xhat)[CS_D], float& rstd, float (&rep)[CS_D]) {
#pragma unroll
    for (int h = 0; h < CS_H; h++) {
        float mx = -INFINITY;
#pragma unroll
        for (int b = 0; b < CS_V; b++) { p[h][b] = T.S[h][a][b] + logc[b]; mx = fmaxf(mx, p[h][b]); }
        float sum = 0.f;
#pragma unroll
        for (int b = 0; b < CS_V; b++) { p[h][b] = __expf(p[h][b] - mx); sum += p[h][b]; }
        const float inv = 1.f / sum;
#pragma unroll
        for (int b = 0; b < CS_V; b++) p[h][b] *= inv;
#pragma unroll
        for (int d = 0; d < CS_HD; d++) {
            float acc = 0.f;
#pragma unroll
            for (int b = 0; b < CS_V; b++) acc += p[h][b] * T.Vt[b][h * CS_HD + d];
            attn[h * CS_HD + d] = acc;
        }
    }
    float o[CS_D], mean = 0.f;
#pragma unroll
    for (int i = 0; i < CS_D; i++) {
        float acc = T.bo[i];
#pragma unroll
        for (int j = 0; j < CS_D; j++) acc += T.W[i][j] * attn[j];
        o[i] = acc; mean += acc;
    }
    mean *= 1.f / CS_D;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < CS_D; i++) { const float c = o[i] - mean; var += c * c; }
    rstd = rsqrtf(var * (1.f / CS_D) + eps);
#pragma unroll
    for (int i = 0; i < CS_D; i++) { xhat[i] = (o[i] - mean) * rstd; rep[i] = xhat[i] * T.lw[i] + T.lb[i]; }
}
// The count vector of a real list is bounded by the deck (game/game.py:77: 14 knights, 5 victory points, 2 year of plenty, 2
// road building, 2 monopoly; id 0 = the reference's placeholder of an empty list): 2 x 15 x 6 x 3 x 3 x 3 = 4 860 possible
// PATTERNS.  The forward kernel can emit each list's pattern number (-1: counts outside the deck, e.g. synthetic test data);
// the backward then sums `dout` per pattern (one index_add) and differentiates 4 860 lists instead of 10^5..10^6.
constexpr int CS_PATTERNS = 2 * 15 * 6 * 3 * 3 * 3;
DEVI int cs_pattern(const float (&cnt)[CS_V]) {
    const int c0 = (int)cnt[0], c1 = (int)cnt[1], c2 = (int)cnt[2], c3 = (int)cnt[3], c4 = (int)cnt[4], c5 = (int)cnt[5];
    if (c0 > 1 || c1 > 14 || c2 > 5 || c3 > 2 || c4 > 2 || c5 > 2) return -1;
    return c0 + 2 * (c1 + 15 * (c2 + 6 * (c3 + 3 * (c4 + 3 * c5))));
}
__global__ __launch_bounds__(256) void k_card_summary_fwd(const void* __restrict__ ids, int esz, long pitch, const int* __restrict__ lens,
                                                          const float* __restrict__ params, float eps, float* __restrict__ out, long rows,
                                                          int* __restrict__ keys) {
    __shared__ CardTables T;
    for (int i = threadIdx.x; i < CS_NPAR; i += 256) reinterpret_cast<float*>(&T)[i] = params[i];
    __syncthreads();
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float cnt[CS_V], logc[CS_V];
    cs_counts(ids, esz, pitch, r, lens[r], cnt);
    if (keys != nullptr) keys[r] = cs_pattern(cnt);
#pragma unroll
    for (int a = 0; a < CS_V; a++) logc[a] = cnt[a] > 0.f ? __logf(cnt[a]) : -INFINITY;
    float acc[CS_D];
#pragma unroll
    for (int i = 0; i < CS_D; i++) acc[i] = 0.f;
    for (int a = 0; a < CS_V; a++) {
        if (cnt[a] == 0.f) continue;
        float p[CS_H][CS_V], attn[CS_D], xhat[CS_D], rep[CS_D], rstd;
        cs_class_fwd(T, logc, a, eps, p, attn, xhat, rstd, rep);
#pragma unroll
        for (int i = 0; i < CS_D; i++) acc[i] += cnt[a] * rep[i];
    }
    float4* o4 = reinterpret_cast<float4*>(out + r * CS_D);
#pragma unroll
    for (int i = 0; i < 4; i++) o4[i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
}
// Inference: the module's output is a function of the list's count PATTERN alone, so a table of the 4 860 patterns' outputs
// (built with k_card_summary_fwd on the synthetic pattern lists whenever the weights change: once per rollout) turns the forward
// into count -> pattern -> one 64-byte row copy (k_card_summary_fwd: ~260 us per 65 536..196 608 lists, a fifth of a policy pass
// at rollout width).  Lists whose counts fall outside the deck (never in a real game) are evaluated directly.
__global__ __launch_bounds__(256) void k_card_summary_lookup(const void* __restrict__ ids, int esz, long pitch, const int* __restrict__ lens,
                                                             const float* __restrict__ table, const float* __restrict__ params, float eps,
                                                             float* __restrict__ out, long rows) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float cnt[CS_V];
    cs_counts(ids, esz, pitch, r, lens[r], cnt);
    const int key = cs_pattern(cnt);
    float4* o4 = reinterpret_cast<float4*>(out + r * CS_D);
    if (key >= 0) {
        const float4* t4 = reinterpret_cast<const float4*>(table + (long)key * CS_D);
#pragma unroll
        for (int i = 0; i < 4; i++) o4[i] = t4[i];
        return;
    }
    const CardTables& T = *reinterpret_cast<const CardTables*>(params);          // (rare: straight from global memory)
    float logc[CS_V], acc[CS_D];
#pragma unroll
    for (int a = 0; a < CS_V; a++) logc[a] = cnt[a] > 0.f ? __logf(cnt[a]) : -INFINITY;
#pragma unroll
    for (int i = 0; i < CS_D; i++) acc[i] = 0.f;
    for (int a = 0; a < CS_V; a++) {
        if (cnt[a] == 0.f) continue;
        float p[CS_H][CS_V], attn[CS_D], xhat[CS_D], rep[CS_D], rstd;
        cs_class_fwd(T, logc, a, eps, p, attn, xhat, rstd, rep);
#pragma unroll
        for (int i = 0; i < CS_D; i++) acc[i] += cnt[a] * rep[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) o4[i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
}
// dparams (CS_NPAR floats, ACCUMULATED into: zero first): gradients of S, V, W, bias, LayerNorm weight / bias.  A parameter
// gradient is a sum over all lists.  Every lane keeps a SLICE of the parameter block in registers over CS_RPL lists, then the
// wave adds its lanes up with shuffles once, one LDS add per wave and one global atomic per block and parameter.  Slices
// (GROUP): 0..3 = four rows of W each (64 accumulators), 4 = V (96), 5 = bias + LayerNorm weight / bias (48), 6 = S (24 per
// query class: the class loop is the outer one there).  Each slice recomputes the (cheap) forward: summing every contribution
// across the wave as it arises cost 2 ms per call, and wider slices spill.
DEVI float cs_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
template <int GROUP>
DEVI void cs_bwd_group(const void* __restrict__ ids, int esz, long pitch, const int* __restrict__ lens,
                       const float* __restrict__ params, float eps, const float* __restrict__ dout,
                       float* __restrict__ dparams, long rows, const int* __restrict__ only_unkeyed, int rpl, int row_blocks) {
    constexpr int NACC = GROUP <= 3 ? 64 : GROUP == 4 ? 96 : GROUP == 5 ? 48 : 24;
    constexpr int OFF_S = 0, OFF_V = CS_H * CS_V * CS_V, OFF_W = OFF_V + CS_V * CS_D, OFF_B = OFF_W + CS_D * CS_D;
    __shared__ CardTables Tshared;
    __shared__ float G[144];
    for (int i = threadIdx.x; i < CS_NPAR; i += 256) reinterpret_cast<float*>(&Tshared)[i] = params[i];
    if (threadIdx.x < 144) G[threadIdx.x] = 0.f;
    __syncthreads();
    const bool lead = (threadIdx.x & 63) == 0;
    // row_blocks > 0 (few rows: the pattern table): the grid is CS_V x row_blocks workgroups and a workgroup takes ONE query class
    // of its rows - with 4 860 lists on 19 x 7 workgroups the launch lasts as long as one lane's chain over the six classes;
    // row_blocks = 0: every lane walks the classes of its lists
    const int bx = row_blocks > 0 ? (int)blockIdx.x % row_blocks : (int)blockIdx.x;
    const int a_only = row_blocks > 0 ? (int)blockIdx.x / row_blocks : -1;
    const long stride = (long)(row_blocks > 0 ? row_blocks : gridDim.x) * 256;
    constexpr int A_OUTER = GROUP == 6 ? CS_V : 1;
#pragma unroll 1
    for (int ao = (GROUP == 6 && a_only >= 0 ? a_only : 0); ao < (GROUP == 6 && a_only >= 0 ? a_only + 1 : A_OUTER); ao++) {
        float acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = 0.f;
#pragma unroll 1
        for (int it = 0; it < rpl; it++) {
            const long r = (long)bx * 256 + threadIdx.x + it * stride;
            if (r >= rows) break;
            if (only_unkeyed != nullptr && only_unkeyed[r] >= 0) continue;      // this list went through its pattern
            float cnt[CS_V], logc[CS_V], dy[CS_D];
            const float4* d4 = reinterpret_cast<const float4*>(dout + r * CS_D);
            bool any = false;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float4 v = d4[i]; dy[4 * i] = v.x; dy[4 * i + 1] = v.y; dy[4 * i + 2] = v.z; dy[4 * i + 3] = v.w;
                any |= v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
            }
            if (!any) continue;                                                 // (a pattern no list of the batch has)
            cs_counts(ids, esz, pitch, r, lens[r], cnt);
#pragma unroll
            for (int a = 0; a < CS_V; a++) logc[a] = cnt[a] > 0.f ? __logf(cnt[a]) : -INFINITY;
#pragma unroll 1
            for (int a = (GROUP == 6 ? ao : (a_only >= 0 ? a_only : 0)); a < (GROUP == 6 ? ao + 1 : (a_only >= 0 ? a_only + 1 : CS_V)); a++) {
                float ca = 0.f;
#pragma unroll
                for (int q = 0; q < CS_V; q++) ca = (q == a) ? cnt[q] : ca;
                if (ca == 0.f) continue;
                // (an opaque zero offset per class: without it the compiler hoists the whole parameter block - 400 LDS values - out of
                //  the loops into registers: every slice compiled to 512 VGPRs, V and S spilled 1.4-1.7 KB, and each reload sat on the lane's serial chain)
                int toff = 0;
                asm volatile("" : "+v"(toff));
                const CardTables& T = *reinterpret_cast<const CardTables*>(reinterpret_cast<const char*>(&Tshared) + toff);
                float p[CS_H][CS_V], attn[CS_D], xhat[CS_D], rep[CS_D], rstd;
                cs_class_fwd(T, logc, a, eps, p, attn, xhat, rstd, rep);
                // LayerNorm backward (drep = count * dy)
                float dxh[CS_D], m1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int i = 0; i < CS_D; i++) {
                    const float dr = ca * dy[i];
                    if constexpr (GROUP == 5) { acc[16 + i] += dr * xhat[i]; acc[32 + i] += dr; }
                    dxh[i] = dr * T.lw[i]; m1 += dxh[i]; m2 += dxh[i] * xhat[i];
                }
                m1 *= 1.f / CS_D; m2 *= 1.f / CS_D;
                float dattn[CS_D];
#pragma unroll
                for (int j = 0; j < CS_D; j++) dattn[j] = 0.f;
#pragma unroll
                for (int i = 0; i < CS_D; i++) {
                    const float d_o = rstd * (dxh[i] - m1 - xhat[i] * m2);           // d(out-projection output i)
                    if constexpr (GROUP == 5) acc[i] += d_o;
#pragma unroll
                    for (int j = 0; j < CS_D; j++) {
                        if constexpr (GROUP <= 3) { if (i / 4 == GROUP) acc[(i % 4) * 16 + j] += d_o * attn[j]; }
                        if constexpr (GROUP >= 4) dattn[j] += d_o * T.W[i][j];
                    }
                }
                if constexpr (GROUP == 4 || GROUP == 6) {
#pragma unroll
                    for (int h = 0; h < CS_H; h++) {
                        float dp[CS_V], dot = 0.f;
#pragma unroll
                        for (int b = 0; b < CS_V; b++) {
                            float s1 = 0.f;
#pragma unroll
                            for (int d = 0; d < CS_HD; d++) {
                                s1 += dattn[h * CS_HD + d] * T.Vt[b][h * CS_HD + d];
                                if constexpr (GROUP == 4) acc[b * CS_D + h * CS_HD + d] += p[h][b] * dattn[h * CS_HD + d];
                            }
                            dp[b] = s1; dot += p[h][b] * s1;
                        }
                        if constexpr (GROUP == 6) {
#pragma unroll
                            for (int b = 0; b < CS_V; b++) acc[h * CS_V + b] += p[h][b] * (dp[b] - dot);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            const float v = cs_wave_sum(acc[i]);
            // slot of accumulator i inside the block's G (group 6: S[h][ao][b] at (h * 6 + ao) * 6 + b)
            const int slot = GROUP == 6 ? ((i / CS_V) * CS_V + ao) * CS_V + (i % CS_V) : i;
            if (lead && v != 0.f) atomicAdd(&G[slot], v);
        }
    }
    __syncthreads();
    constexpr int NOUT = GROUP == 6 ? 144 : NACC;
    constexpr int BASE = GROUP <= 3 ? OFF_W + 64 * GROUP : GROUP == 4 ? OFF_V : GROUP == 5 ? OFF_B : OFF_S;
    if (threadIdx.x < NOUT && G[threadIdx.x] != 0.f) atomicAdd(&dparams[BASE + threadIdx.x], G[threadIdx.x]);
}
// blockIdx.y = the parameter slice: the seven slices of one backward run side by side in ONE launch
__global__ __launch_bounds__(256) void k_card_summary_bwd(const void* __restrict__ ids, int esz, long pitch, const int* __restrict__ lens,
                                                          const float* __restrict__ params, float eps, const float* __restrict__ dout,
                                                          float* __restrict__ dparams, long rows, const int* __restrict__ only_unkeyed, int rpl,
                                                          const int* __restrict__ n_unkeyed, int row_blocks) {
    if (n_unkeyed != nullptr && *n_unkeyed == 0) return;      // (uniform) every list went through its pattern
    switch (blockIdx.y) {
    case 0: cs_bwd_group<0>(ids, esz, pitch, lens, params, eps, dout, dparams, rows, only_unkeyed, rpl, row_blocks); break;
    case 1: cs_bwd_group<1>(ids, esz, pitch, lens, params, eps, dout, dparams, rows, only_unkeyed, rpl, row_blocks); break;
    case 2: cs_bwd_group<2>(ids, esz, pitch, lens, params, eps, dout, dparams, rows, only_unkeyed, rpl, row_blocks); break;
    case 3: cs_bwd_group<3>(ids, esz, pitch, lens, params, eps, dout, dparams, rows, only_unkeyed, rpl, row_blocks); break;
    case 4: cs_bwd_group<4>(ids, esz, pitch, lens, params, eps, dout, dparams, rows, only_unkeyed, rpl, row_blocks); break;
    case 5: cs_bwd_group<5>(ids, esz, pitch, lens, params, eps, dout, dparams, rows, only_unkeyed, rpl, row_blocks); break;
    default: cs_bwd_group<6>(ids, esz, pitch, lens, params, eps, dout, dparams, rows, only_unkeyed, rpl, row_blocks); break;
    }
}

// dpat[key[r]] += dout[r] for the rows with key >= 0.  Many rows share a pattern (a minibatch of 614 400 lists holds ~1 400
// distinct ones, a handful of them cover most lists), so neither plain global atomics (they pile up on a few cache lines) nor
// adding up the lanes of a wave per distinct pattern (one shuffle tree of 16 values per distinct key in the wave: 526 us) fit.
// A workgroup sums its 1 024 rows in an LDS hash table first (512 slots of key + 16 floats, native ds_add_f32; a row whose 16
// probes all hit other keys adds to global memory directly) and then adds each occupied slot to the table once.
// dpat holds `replicas` copies of the table ([replicas][CS_PATTERNS][16], summed by the caller): workgroup b adds to copy
// b % replicas.  n_unkeyed (may be null) += the rows with key < 0: the caller's per-row backward for those returns at once
// when there are none (the usual case: their counts are outside the deck).
constexpr int CPS_SLOTS = 512, CPS_PITCH = 17, CPS_ROWS = 1024;
__global__ __launch_bounds__(256) void k_card_pattern_sum(const int* __restrict__ keys, const float* __restrict__ dout, float* __restrict__ dpat, long rows,
                                                          int replicas, int* __restrict__ n_unkeyed) {
    __shared__ int skey[CPS_SLOTS];
    __shared__ float sval[CPS_SLOTS * CPS_PITCH];
    dpat += (long)(blockIdx.x % replicas) * CS_PATTERNS * CS_D;
    for (int i = threadIdx.x; i < CPS_SLOTS; i += 256) skey[i] = -1;
    for (int i = threadIdx.x; i < CPS_SLOTS * CPS_PITCH; i += 256) sval[i] = 0.f;
    __syncthreads();
    int unk = 0;
#pragma unroll 1
    for (int it = 0; it < CPS_ROWS / 256; it++) {
        const long r = (long)blockIdx.x * CPS_ROWS + it * 256 + threadIdx.x;
        if (r >= rows) break;
        const int key = keys[r];
        if (key < 0) { unk++; continue; }
        float v[CS_D];
        const float4* d4 = reinterpret_cast<const float4*>(dout + r * CS_D);
#pragma unroll
        for (int i = 0; i < 4; i++) { const float4 q = d4[i]; v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }
        int s = (int)(((unsigned)key * 2654435761u) >> 23), slot = -1;
#pragma unroll 1
        for (int probe = 0; probe < 16; probe++, s = (s + 1) & (CPS_SLOTS - 1)) {
            const int old = atomicCAS(&skey[s], -1, key);
            if (old == -1 || old == key) { slot = s; break; }
        }
        if (slot >= 0) {
#pragma unroll
            for (int i = 0; i < CS_D; i++) atomicAdd(&sval[slot * CPS_PITCH + i], v[i]);
        } else {
#pragma unroll
            for (int i = 0; i < CS_D; i++) atomicAdd(&dpat[(long)key * CS_D + i], v[i]);
        }
    }
    if (n_unkeyed != nullptr && unk) atomicAdd(n_unkeyed, unk);
    __syncthreads();
    for (int e = threadIdx.x; e < CPS_SLOTS * CS_D; e += 256) {
        const int slot = e / CS_D, i = e % CS_D, key = skey[slot];
        if (key < 0) continue;
        const float x = sval[slot * CPS_PITCH + i];
        if (x != 0.f) atomicAdd(&dpat[(long)key * CS_D + i], x);
    }
}

}  // namespace catan
