// catan_wgrad_big.hip - the weight gradient of a WIDE linear layer over many rows: dW [O][I] = dY^T X with dY [R][O], X [R][I], R ~ 2 x 10^5,
// O, I in the hundreds - the observation trunk's 992 -> 512 product of a PPO minibatch (RL/models/observation_module.py:58-63 under
// RL/ppo/ppo.py:66's backward): 208 GFLOP over 616 MB of operands, the one GEMM-shaped weight gradient of the net whose FLOPs matter
// (0.083 ms at the dense bf16 MFMA peak, 0.1 ms at the HBM rate; the library's split-K kernel takes 0.62 ms).
//
// The k_wgrad_tr scheme (rows split over the grid, operands staged through LDS in transposing sub-tile images, fp32 accumulators in
// registers for the whole slice) holds one workgroup's output in 128 VGPRs: a 128 x 128 tile.  What makes that tile size affordable is
// WHO shares the operand bytes.  A slice of 64 rows is 96 KB of X and 64 KB of dY, and every one of the 32 output tiles needs a
// quarter / an eighth of it: the 32 workgroups that work on the SAME rows are placed in the same XCD (the dispatcher deals workgroups
// to the 8 XCDs round-robin: workgroup b runs on XCD b % 8, so tile = (b / 8) % 32, row group = b % 8 + 8 (b / 256)) and walk their
// rows in step - each row slice comes from HBM once into that XCD's L2 and is read from there by the 32 CUs.  HBM sees the operands
// once; the L2 serves 32 KB per workgroup and stage.
//
// Every (row group, tile) writes its fp32 partial tile once (no atomics); k_wgrad_big_reduce adds the row groups' partials in index
// order into dW / db: the result does not depend on which workgroup finished first.  Column I of the X image is all ones (db for free).
#pragma once

namespace catan {

constexpr int WB_T = 128;                                    // output tile: 128 outputs x 128 inputs per workgroup
constexpr int WB_IMG = (WB_T / 16) * (WG_KT / 32) * WG_SUB;  // elements of a 64-row x 128-column operand image

// part: float [groups][O][IP] (IP = tiles_i * 128); rows of group g: [g * rows_per_group, ...)
__global__ __launch_bounds__(256) void k_wgrad_big(const unsigned short* __restrict__ X, const unsigned short* __restrict__ dY, float* __restrict__ part,
                                                   long R, int I, int O, long rows_per_group, int tiles_o, int tiles_i) {
    __shared__ __attribute__((aligned(16))) unsigned short Xs[WB_IMG];
    __shared__ __attribute__((aligned(16))) unsigned short Ys[WB_IMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wo = wave >> 1, wi = wave & 1;
    const int tiles = tiles_o * tiles_i;
    const long b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const long q = b >> 3;
    const int tile = (int)(q % tiles), grp = xcd + 8 * (int)(q / tiles);
    const int to = tile % tiles_o, ti = tile / tiles_o;
    const long r_begin = (long)grp * rows_per_group;
    const long r_end = r_begin + rows_per_group < R ? r_begin + rows_per_group : R;
    const int IP = tiles_i * WB_T;
    const int xc0 = ti * WB_T;
    const int xw = I - xc0 < WB_T ? (I - xc0 > 0 ? I - xc0 : 0) : WB_T;      // valid X columns of this tile (a multiple of 8)
    const int ones = (I >= xc0 && I < xc0 + WB_T) ? I - xc0 : -1;            // the column of ones (bias gradient), if it falls into this tile
    for (int x = tid; x < WB_IMG; x += 256) { Xs[x] = 0; Ys[x] = 0; }
    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[a][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    uint4 vx[4], vy[4];
    if (r_begin < r_end) {
        if (xw > 0) wg_load_cols<4>(X, r_begin, R, (long)I, xc0, xw, vx, tid);
        wg_load_cols<4>(dY, r_begin, R, (long)O, to * WB_T, WB_T, vy, tid);
    }
    __syncthreads();                                          // zero fill complete
    for (long r0 = r_begin; r0 < r_end; r0 += WG_KT) {
        if (xw > 0) wg_store_rows<4>(Xs, xw, vx, tid);
        wg_store_rows<4>(Ys, WB_T, vy, tid);
        if (ones >= 0 && tid < WG_KT) Xs[wg_sub_off(ones >> 4, tid >> 5) + (tid & 31) * 16 + (ones & 15)] = (r0 + tid < r_end) ? (unsigned short)0x3F80 : (unsigned short)0;
        __syncthreads();
        if (r0 + WG_KT < r_end) {                             // the next stage's rows fly during the MFMAs
            if (xw > 0) wg_load_cols<4>(X, r0 + WG_KT, R, (long)I, xc0, xw, vx, tid);
            wg_load_cols<4>(dY, r0 + WG_KT, R, (long)O, to * WB_T, WB_T, vy, tid);
        }
#pragma unroll
        for (int ks = 0; ks < WG_KT / 32; ks++) {
            bf16x8_t bf[4];
#pragma unroll
            for (int c = 0; c < 4; c++) bf[c] = wg_frag_tr(Xs, wi * 4 + c, ks, lane);
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const bf16x8_t af = wg_frag_tr(Ys, wo * 4 + a, ks, lane);
#pragma unroll
                for (int c = 0; c < 4; c++) acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[c], acc[a][c], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* P = part + ((long)grp * O) * IP;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int o = to * WB_T + (wo * 4 + a) * 16 + 4 * (lane >> 4) + r, i = xc0 + (wi * 4 + c) * 16 + (lane & 15);
                P[(long)o * IP + i] = acc[a][c][r];
            }
}

// dw [O][ldw] (+)= sum over the groups (in index order) of part[g][o][i], i < I; db [O] (+)= the ones column.  accumulate = 0: overwrite.
__global__ __launch_bounds__(256) void k_wgrad_big_reduce(const float* __restrict__ part, int groups, int O, int I, int IP, float* __restrict__ dw, long ldw,
                                                          float* __restrict__ db, int accumulate) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)O * (I + 1)) return;
    const int o = (int)(idx / (I + 1)), i = (int)(idx - (long)o * (I + 1));
    float s = 0.f;
    for (int g = 0; g < groups; g++) s += part[((long)g * O + o) * IP + i];
    if (i < I) { float* d = dw + (long)o * ldw + i; *d = accumulate ? *d + s : s; }
    else if (db != nullptr) db[o] = accumulate ? db[o] + s : s;
}

}  // namespace catan
