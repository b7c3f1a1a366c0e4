/* sbr_wave.hip — the recurrent pass of SMALL minibatches at d <= 32: one wavefront per sequence.
 *
 * The reference's own schedule is one optimiser step per subsequence (sequence_model.rs:111-169) at embedding_dim 32
 * (lib.rs:22-58): a step is then ONE chain of up to max_sequence_length dependent LSTM cells in each direction.  On the
 * MFMA tile kernels (sbr_kernels.hip) such a step costs a full 16-row tile per time step — 15 of the 16 rows are
 * padding — and two barriers; here the recurrence runs on the vector ALU of a single wave, with no barrier at all:
 *
 *   forward   z_t = bW + x_t Wx + h_{t-1} Wh is a k-ascending fma chain whose x part comes first, so
 *             P_t = bW + x_t Wx does not depend on the recurrence: all 256 threads of the workgroup compute P_t for every
 *             t into LDS up front (and copy the gathered rows to X for the dense-gradient GEMM); then wave 0 walks t:
 *             lane l continues the chains of gate columns l and l + 64 over h_{t-1}.  d = 32 with four gates (the
 *             reference's configuration): one packed fma per k with h_k from a v_readlane, the rational tanh of the lane's
 *             own two pre-activations, numerators / denominators across the half-waves by v_permlane32_swap — no LDS
 *             between two steps.  Other shapes: h_{t-1} and the pre-activations cross lanes through LDS, every lane
 *             evaluates the cell of unit l mod d.  The lane groups share the stores of G / C / H.
 *   backward  the gate values, cell states, tanh(c_t) and dloss/dh rows of every step are staged into LDS by all 256
 *             threads (the transcendental of the cell backward does not depend on the gradient coming down the sequence);
 *             wave 0 walks t downwards: cell backward in lane u = l mod d, dz to LDS and HBM, then lane c owns column c
 *             of dz W^T (row c of W in registers: 4d values) — columns < d are dX of the row, columns >= d the recurrent
 *             dh, which returns to the unit lanes by v_permlane32_swap / v_permlane16_swap.
 *   also here lstm_dw_block_kernel: the dense gradient of such steps with one wave per 32 x 32 output block.
 *
 * Arithmetic is the contract's, operation for operation (sbr_numerics.h; fma chains in k / j order from the bias / from
 * 0), so the results are bit-identical to the tile kernels' and the oracle's; tests/test_parity_gpu.py runs both forms.
 * A workgroup per sequence, any max_sequence_length (segments of SBR_WAVE_SEG steps are staged in LDS at a time): up to
 * SBR_WAVE_MAX_SEQ sequences per step (twice that at d = 16) take this path (launch_wave_*
 * return false otherwise and the caller launches the tile kernels).  The machine model behind the design — a lone wave
 * issues a dependent vector instruction every 8.8 cycles, so a step costs its instruction count — is measured by
 * tools/valu_chain_ubench.hip; numbers in profiles/r03_small_steps.md. */
#include <atomic>
#include <cstdlib>

#include "sbr_kernels.h"
#include "sbr_numerics.h"
#include "sbr_wave_seq.h"

namespace sbr {
namespace {

#ifndef SBR_WAVE_MAX_SEQ
/* ms per step, len <= 64, wave / tile form of the recurrent kernels (everything else equal): d = 32: 1 sequence 0.126 / 0.216,
 * 256 0.224 / 0.437, 2 048 0.336 / 0.458, 4 096 0.629 / 0.603, 8 192 1.01 / 0.705; d = 16: 2 048 0.230 / 0.373, 4 096 0.411 / 0.477,
 * 8 192 0.501 / 0.506 — the wave form up to 2 048 sequences per step at d = 32, 4 096 at d = 16 */
#define SBR_WAVE_MAX_SEQ 2048
#endif
constexpr size_t WAVE_LDS_LIMIT = 150 * 1024;
#ifndef SBR_WAVE_SEG
#define SBR_WAVE_SEG 128 /* time steps of a sequence staged in LDS at a time (80 / 112 KB forward / backward at d = 32) */
#endif
#define SBR_WAVE_RSRC_FLAGS 0x00020000 /* word 3 of a raw buffer resource on gfx9-class targets (32-bit data format, no swizzle) */
#ifndef SBR_DW_BLOCK_MAX_CHUNKS
#define SBR_DW_BLOCK_MAX_CHUNKS 128 /* x 1-2 block rows = workgroups: the chip's CUs once */
#endif

using namespace wave_seq;

// Both kernels walk a sequence in SEGMENTS of at most `seg` time steps (SBR_WAVE_SEG: what of a sequence is staged in LDS at
// a time — any max_sequence_length fits); the bodies are in sbr_wave_seq.h.
template <int D, int NG>
__global__ __launch_bounds__(256) void lstm_fwd_wave_kernel(ModelView m, MbView mb, float* H, WorkView w, int seg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lstm_fwd_wave_seq<D, NG>(m, mb, H, w, seg, (int)blockIdx.x, lds);
}

template <int D, int NG>
__global__ __launch_bounds__(256) void lstm_bwd_wave_kernel(ModelView m, MbView mb, BlockView blk, WorkView w, int seg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    lstm_bwd_wave_seq<D, NG>(m, mb, blk, w, seg, (int)blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------------
// Dense gradient of a SMALL step: dW[k][j] = sum_r xh[r][k] dz[r][j] per 1 024-row chunk, one WAVE per 32 x 32 block of
// the output.  The chunk's chain (rows ascending, from 0: SBR_DW_CHUNK_ROWS) is 512 dependent v_mfma_f32_32x32x2_f32 for
// every accumulator, whoever computes it; the tile kernel of sbr_kernels.hip gives a wave four accumulators (and at
// d = 32 half of its 128 x 128 tile lies outside the matrix), so a chunk costs it 4 x 512 MFMAs back to back — 55-130 us
// — which is what a step of a few hundred sequences then waits for.  Here a chunk is 512 MFMAs deep (14 us): one
// workgroup per chunk and block ROW, one wave per block — the waves of a workgroup sit on different SIMDs, each with an
// MFMA pipe to itself (two dependent chains on one SIMD take turns: measured 141 cycles per MFMA and wave) — 64-row slabs of
// [32 columns of xh | dz] staged through LDS (double-buffered in LDS, two slabs ahead in registers), operands read back
// one float per lane and MFMA, eight MFMAs' reads ahead.  Rows past the chunk's end, first-step rows'
// h_{t-1} and padding columns are zeros (requests beyond the buffer resources), i.e. +0 on every chain.  The bias row
// (column sums of dz, a plain add chain over the rows) is formed by the waves of the first block row from the staged dz.
// ------------------------------------------------------------------------------------------------
typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef unsigned v4uw __attribute__((ext_vector_type(4)));
template <int D, int NG>
struct DwBlockCfg {
    static constexpr int K2 = 2 * D, NGD = NG * D;
    static constexpr int KB = (K2 + 31) / 32, JB = (NGD + 31) / 32;  // blocks; a workgroup = one block ROW of one chunk
    static constexpr int NL = 4;                                     // loader waves (one per SIMD, beside the MFMA waves)
    static constexpr int NT = (JB + NL + 1) * 64;                    // + the wave that forms the bias row
    static constexpr int KP = 32, JP = JB * 32, LD = KP + JP;        // floats per staged row: the block row's 32 columns of xh, all of dz
    static constexpr int S = 64;                                     // rows per slab
    static constexpr size_t lds_bytes = (size_t)2 * S * LD * 4 + SBR_DW_CHUNK_ROWS * 4;
};
template <int D, int NG>
__global__ __launch_bounds__((DwBlockCfg<D, NG>::NT)) void lstm_dw_block_kernel(ModelView m, MbView mb, BlockView blk, WorkView w) {
    using Cfg = DwBlockCfg<D, NG>;
    constexpr int K2 = Cfg::K2, NGD = Cfg::NGD, JB = Cfg::JB, NT = Cfg::NT, KP = Cfg::KP, LD = Cfg::LD, S = Cfg::S;
    constexpr int NLT = Cfg::NL * 64;                         // loader threads
    constexpr int QX = 32 / 4, QZ = NGD / 4;                  // 16-byte pieces per row of the xh block row and of dZ
    constexpr int IX = (S * QX + NLT - 1) / NLT, IZ = (S * QZ + NLT - 1) / NLT;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* buf0 = lds;
    float* buf1 = lds + S * LD;
    int* s_prev = reinterpret_cast<int*>(lds + 2 * S * LD);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = blockIdx.x % Cfg::KB;
    const int c = blockIdx.x / Cfg::KB;
    const int r0 = c * SBR_DW_CHUNK_ROWS;
    const int nr = mb.R - r0 < SBR_DW_CHUNK_ROWS ? mb.R - r0 : SBR_DW_CHUNK_ROWS;
    const int nslabs = (nr + S - 1) / S;
    for (int i = tid; i < SBR_DW_CHUNK_ROWS; i += NT) s_prev[i] = i < nr ? mb.prev_row[r0 + i] : -1;
    // padding columns of both buffers (d = 16 with three gates: dz columns 48..63) stay zero: cleared once
    for (int i = tid; i < 2 * S * LD; i += NT) lds[i] = 0.0f;
    __syncthreads();
    float* part = w.partials + (size_t)c * (K2 + 1) * NGD;
    if (wave < JB) {
        // ---- MFMA wave: block (kb, jb = wave); per slab 32 dependent MFMAs, operands from LDS, eight MFMAs' reads ahead
        const int jb = wave, l31 = lane & 31, par = lane >> 5;
        f32x16w acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
        __syncthreads();  // slab 0 staged
        for (int slab = 0; slab < nslabs; ++slab) {
            const float* cur = (slab & 1) ? buf1 : buf0;
            const float* ap = cur + par * LD + l31;
            const float* bp = cur + par * LD + KP + jb * 32 + l31;
            constexpr int G = 8;
            float av[2][G], bv[2][G];
#pragma unroll
            for (int q = 0; q < G; ++q) { av[0][q] = ap[2 * q * LD]; bv[0][q] = bp[2 * q * LD]; }
#pragma unroll
            for (int g = 0; g < S / 2 / G; ++g) {
                if (g + 1 < S / 2 / G) {
#pragma unroll
                    for (int q = 0; q < G; ++q) {
                        av[(g + 1) & 1][q] = ap[2 * ((g + 1) * G + q) * LD];
                        bv[(g + 1) & 1][q] = bp[2 * ((g + 1) * G + q) * LD];
                    }
                }
#pragma unroll
                for (int q = 0; q < G; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][q], bv[g & 1][q], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
        const int j = jb * 32 + l31;
        if (j < NGD) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int kk = kb * 32 + (q & 3) + 8 * (q >> 2) + 4 * par;
                if (kk < K2) part[(size_t)kk * NGD + j] = acc[q];
            }
        }
        return;
    }
    if (wave == JB + Cfg::NL) {
        // ---- bias wave: row k = 2d of the gradient = column sums of dz, a plain add chain over the rows of the chunk (rows past
        // its end add +0), read from the staged slabs; lane l owns columns l and 64 + l.  (A lone wave pays ~7 cycles per
        // instruction: folded into the MFMA waves these adds cost a third of the MFMA rate, folded into a loader wave they made
        // it the slowest wave of the workgroup.)  Only the first block row's workgroup has work here.
        float bias0 = 0.0f, bias1 = 0.0f;
        __syncthreads();  // slab 0 staged
        for (int slab = 0; slab < nslabs; ++slab) {
            if (kb == 0) {
                const float* cur = (slab & 1) ? buf1 : buf0;
                const float* z0 = cur + KP + (lane < NGD ? lane : 0);
                const float* z1 = cur + KP + (64 + lane < NGD ? 64 + lane : 0);
#pragma unroll 16
                for (int i = 0; i < S; ++i) {
                    bias0 = bias0 + z0[i * LD];
                    bias1 = bias1 + z1[i * LD];
                }
            }
            __syncthreads();
        }
        if (kb == 0) {
            if (lane < NGD) part[(size_t)K2 * NGD + lane] = bias0;
            if (64 + lane < NGD) part[(size_t)K2 * NGD + 64 + lane] = bias1;
        }
        return;
    }
    // ---- loader waves: stage the slabs (rows past the chunk, h_{t-1} of first steps, columns outside the matrix: requests beyond
    // the buffer resources, i.e. zeros)
    const int lt = tid - JB * 64;  // loader thread id
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)w.X, (short)0, mb.R * D * 4, SBR_WAVE_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void*)blk.H, (short)0, mb.R * D * 4, SBR_WAVE_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void*)w.dZ, (short)0, mb.R * NGD * 4, SBR_WAVE_RSRC_FLAGS);
    struct Regs { v4uw xr[IX], hr[IX], zr[IZ]; };
    Regs ra, rb;  // two slabs in flight towards LDS (a request has two slabs' MFMAs to land under)
    auto fetch = [&](Regs& rg, int slab) {
#pragma unroll
        for (int it = 0; it < IX; ++it) {
            const int p = lt + it * NLT, lr = slab * S + p / QX, k4 = kb * 32 + (p % QX) * 4;  // column of xh = [x | h]
            const bool ok = p < S * QX && lr < nr;
            const int pr = s_prev[lr < SBR_DW_CHUNK_ROWS ? lr : 0];
            // one of the two requests of a piece falls outside its resource and returns zeros: the piece is their OR
            rg.xr[it] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok && k4 < D ? (unsigned)((r0 + lr) * D + k4) * 4u : OOB, 0, 0);
            rg.hr[it] = __builtin_amdgcn_raw_buffer_load_b128(rsH, ok && k4 >= D && k4 < K2 && pr >= 0 ? (unsigned)(pr * D + k4 - D) * 4u : OOB, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < IZ; ++it) {
            const int p = lt + it * NLT, lr = slab * S + p / QZ, c4 = (p % QZ) * 4;
            const bool ok = p < S * QZ && lr < nr;
            rg.zr[it] = __builtin_amdgcn_raw_buffer_load_b128(rsZ, ok ? (unsigned)((r0 + lr) * NGD + c4) * 4u : OOB, 0, 0);
        }
    };
    auto stage = [&](const Regs& rg, float* buf) {
#pragma unroll
        for (int it = 0; it < IX; ++it) {
            const int p = lt + it * NLT, row = p / QX, c4 = (p % QX) * 4;
            if (p < S * QX) *reinterpret_cast<v4uw*>(&buf[row * LD + c4]) = rg.xr[it] | rg.hr[it];
        }
#pragma unroll
        for (int it = 0; it < IZ; ++it) {
            const int p = lt + it * NLT, row = p / QZ, c4 = (p % QZ) * 4;
            if (p < S * QZ) *reinterpret_cast<v4uw*>(&buf[row * LD + KP + c4]) = rg.zr[it];
        }
    };
    // slab s is computed from LDS buffer s & 1 while slab s + 1 goes from registers into the other buffer (its readers passed
    // the barrier at the end of slab s - 1) and slab s + 3 is requested into the registers that have just been emptied;
    // requests past the last slab fall outside the chunk and return zeros, which nobody reads
    fetch(ra, 0);
    stage(ra, buf0);
    fetch(ra, 1);
    fetch(rb, 2);
    __syncthreads();  // slab 0 staged
    for (int slab = 0; slab < nslabs; slab += 2) {
        stage(ra, buf1);
        fetch(ra, slab + 3);
        __syncthreads();
        if (slab + 1 < nslabs) {
            stage(rb, buf0);
            fetch(rb, slab + 4);
            __syncthreads();
        }
    }
}

int wave_mode() {  // SBR_WAVE: 0 never, 1 whenever the shape allows, unset: up to SBR_WAVE_MAX_SEQ sequences per step
    const char* e = std::getenv("SBR_WAVE");  // read per call: the tests force either form
    return e ? (std::atoi(e) ? 1 : 0) : -1;
}
bool wave_shape_ok(int d, int ng, int B, size_t lds_bytes) {
    if ((d != 16 && d != 32) || (ng != 3 && ng != 4) || B <= 0 || lds_bytes > WAVE_LDS_LIMIT) return false;
    const int mode = wave_mode();
    return mode == 1 || (mode < 0 && B <= (d == 16 ? 2 * SBR_WAVE_MAX_SEQ : SBR_WAVE_MAX_SEQ));
}
// dynamic LDS beyond the default limit has to be granted per kernel; `granted` is the instantiation's high-water mark, so a
// step of a few microseconds does not pay for the call again
// (per device: the attribute belongs to the function on the current device, and one process may drive several)
template <class K>
void allow_lds(K kernel, size_t bytes, std::atomic<size_t>* granted) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev >= 0 && dev < 64 ? dev : 0;
    if (bytes <= granted[dev]) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    granted[dev] = bytes;
}

}  // namespace

bool launch_wave_forward(const ModelView& m, const MbView& mb, float* H, const WorkView& w, int tm_host, hipStream_t s) {
    const int d = m.d, ngd = m.ng * m.d;
    const int seg = tm_host < SBR_WAVE_SEG ? tm_host : SBR_WAVE_SEG;
    const size_t lds = ((size_t)seg * (d + ngd) + ngd + d + seg) * 4;
    if (!wave_shape_ok(d, m.ng, mb.B, lds)) return false;
#define SBR_WAVE_FWD(DD, NN)                                                                                   \
    {                                                                                                          \
        static std::atomic<size_t> granted[64];                                                                             \
        allow_lds(lstm_fwd_wave_kernel<DD, NN>, lds, granted);                                                \
        hipLaunchKernelGGL((lstm_fwd_wave_kernel<DD, NN>), dim3(mb.B), dim3(256), lds, s, m, mb, H, w, seg);           \
    }
    if (d == 32 && m.ng == 4) SBR_WAVE_FWD(32, 4)
    else if (d == 32) SBR_WAVE_FWD(32, 3)
    else if (m.ng == 4) SBR_WAVE_FWD(16, 4)
    else SBR_WAVE_FWD(16, 3)
#undef SBR_WAVE_FWD
    return true;
}

bool launch_wave_backward(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int tm_host, int b_host,
                          hipStream_t s) {
    const int d = m.d, ngd = m.ng * m.d;
    const int seg = tm_host < SBR_WAVE_SEG ? tm_host : SBR_WAVE_SEG;
    const size_t lds = ((size_t)seg * 7 * d + d + ngd + seg) * 4;
    if (!wave_shape_ok(d, m.ng, b_host, lds)) return false;
#define SBR_WAVE_BWD(DD, NN)                                                                                   \
    {                                                                                                          \
        static std::atomic<size_t> granted[64];                                                                             \
        allow_lds(lstm_bwd_wave_kernel<DD, NN>, lds, granted);                                                \
        hipLaunchKernelGGL((lstm_bwd_wave_kernel<DD, NN>), dim3(b_host), dim3(256), lds, s, m, mb, blk, w, seg);       \
    }
    if (d == 32 && m.ng == 4) SBR_WAVE_BWD(32, 4)
    else if (d == 32) SBR_WAVE_BWD(32, 3)
    else if (m.ng == 4) SBR_WAVE_BWD(16, 4)
    else SBR_WAVE_BWD(16, 3)
#undef SBR_WAVE_BWD
    return true;
}

bool launch_wave_dense_gradient(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int rows_host,
                                hipStream_t s) {
    const int d = m.d;
    if ((d != 16 && d != 32) || (m.ng != 3 && m.ng != 4) || rows_host <= 0) return false;
    const char* e = std::getenv("SBR_DW_BLOCK");  // 0 never, 1 always (tests), unset: while the waves fit the SIMDs about once
    const int mode = e ? (std::atoi(e) ? 1 : 0) : -1;
    const int nch = (rows_host + SBR_DW_CHUNK_ROWS - 1) / SBR_DW_CHUNK_ROWS;
    if (mode == 0 || (mode < 0 && nch > SBR_DW_BLOCK_MAX_CHUNKS)) return false;
    if ((size_t)rows_host * m.ng * d * 4 >= ((size_t)1 << 31)) return false; /* the kernel's 32-bit buffer offsets */
#define SBR_DW_BLOCK_LAUNCH(DD, NN)                                                                                          \
    {                                                                                                                        \
        using Cfg = DwBlockCfg<DD, NN>;                                                                                      \
        static std::atomic<size_t> granted[64];                                                                                           \
        allow_lds(lstm_dw_block_kernel<DD, NN>, Cfg::lds_bytes, granted);                                                    \
        hipLaunchKernelGGL((lstm_dw_block_kernel<DD, NN>), dim3(nch * Cfg::KB), dim3(Cfg::NT), Cfg::lds_bytes, s, m, mb, blk, w); \
    }
    if (d == 32 && m.ng == 4) SBR_DW_BLOCK_LAUNCH(32, 4)
    else if (d == 32) SBR_DW_BLOCK_LAUNCH(32, 3)
    else if (m.ng == 4) SBR_DW_BLOCK_LAUNCH(16, 4)
    else SBR_DW_BLOCK_LAUNCH(16, 3)
#undef SBR_DW_BLOCK_LAUNCH
    return true;
}

}  // namespace sbr
