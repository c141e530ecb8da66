// 1x1 conv (fp16) with the PIXEL operand loaded per lane straight into the MFMA B fragment -- no LDS for it.
//
// Why: the 1x1 layers of Darknet-53 at 38 x 38 / 19 x 19 are ONE round of workgroups whose K loop streams both operands through
// LDS-DMA rings; the loop is bound by bytes in flight x round-trip latency, and the bytes in flight are bounded by the 160 KiB of
// LDS (conv_dma.hip: ~50 GB/s per CU whatever the tile; profiles/r03_ablation.md).  For a 1x1 conv the B fragment of
// mfma_f32_16x16x32_f16 -- lane (fr, fq): pixel fr, channels 8 fq .. 8 fq + 7 of the 32-deep step -- is 16 CONTIGUOUS bytes of the
// NHWC input, so a lane can `buffer_load_dwordx4` it directly (range check = M tail, as in the LDS-DMA kernels).  Then only the
// weights need the ring, the pixel bytes in flight live in registers, and a CU holds ~100 KiB of pixels + 64 KiB of weights in
// flight instead of 96 KiB in all.
//
// Tile: 128 couts x 256 pixels, eight waves as 1 x 8: every wave owns ALL 128 couts (TM = 8) of its own 32 pixels (TP = 2), so no two
// waves load the same pixel bytes.  K in super-steps of 64 channels (two MFMA k-steps): per super-step a lane loads 2 x 2 x 16 bytes
// (two pixels x two halves of the 128-byte line of its pixel -- the four lanes of a pixel cover the line), the weights arrive as a
// 128-row x 128-byte stage (16 KiB) of a four-slot LDS-DMA ring, requested three super-steps ahead.  The B registers of k-step s are
// re-requested for super-step kt + 2 right behind the MFMAs that consumed them (two register sets: 32 VGPRs at the 128-register
// limit of two workgroups per CU), i.e. ~1.5 super-steps ahead.  vmcnt is in order over both kinds of request: every iteration
// issues 2 weight DMAs + 4 pixel loads, and the wait at the top of iteration kt leaves exactly the previous iteration's six in flight.
#include "conv_common.h"
#include <type_traits>

namespace yolo {

namespace {

typedef __attribute__((address_space(3))) void x1_lds_void;

__device__ __forceinline__ void x1_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char *lds_dst, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (x1_lds_void *)lds_dst, 16, voff, soff, 0, 0);
#else
    (void)rsrc; (void)lds_dst; (void)voff; (void)soff;
#endif
}

template <int N>
__device__ __forceinline__ void x1_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// EPI: 0 the generic epilogue (any output map / view), 1 the lean one (plain fp16 output map) -- instantiations, not branches
template <int EPI>
__global__ void __launch_bounds__(512, 4) conv1x1_regb_kernel(const ConvParams p) {
    typedef _Float16 T;
    constexpr int TM = 8, TP = 2, NA = 128, NB = 256, S = 4, ROWB = 128, CH = 4 * TM;
    constexpr int A_BYTES = NA * ROWB;          // one weight stage: 128 couts x 64 channels
    constexpr int JA = 2;                       // weight DMA wave-instructions per wave and stage (16 x 1 KiB over 8 waves)
    constexpr int NI = JA + 2 * TP;             // vector-memory requests per wave and iteration
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * A_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    const int bid = xcd_remap(blockIdx.x, p.n_blocks);
    const int mt = (int)fdiv((uint32_t)bid, p.dtiles_n);
    const int nt = bid - mt * p.n_tiles_n;
    const int n0 = nt * NA;
    const int m0 = mt * NB;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    // ---- weight DMA geometry (as conv_dma.hip, 128-byte rows): wave instruction j fills rows 8 (8 j + wave) .. + 7, lane -> row
    // lane / 8, physical chunk lane % 8 = logical chunk ^ ((row >> 1) & 7); LDS row r holds the cout that makes a lane own CH
    // contiguous couts (conv_common.h: conv_epilogue)
    const int lrow = lane >> 3;
    const uint32_t csw = (uint32_t)(((lane & 7) ^ ((4 * (wave & 1) + (lrow >> 1)) & 7)) << 4);
    uint32_t a_off[JA];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int r = (j * 8 + wave) * 8 + lrow;
        const int tm = r >> 4, g4 = (r >> 2) & 3, jj = r & 3;
        const int ch = g4 * CH + 4 * tm + jj;
        a_off[j] = (uint32_t)(n0 + ch) * p.wrow_bytes + csw;
    }
    // ---- pixel operand: this lane's two pixels, 16 bytes at channel chunk fq of every 32-deep k-step
    uint32_t b_off[TP];
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        const int m = m0 + wave * (TP * 16) + b * 16 + fr;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = (int)fdiv((uint32_t)mm, p.dHoWo);
        const int rem = mm - n * p.HoWo;
        const long long e = (long long)n * p.in_img_stride + (long long)rem * p.in_ld + p.in_coff;
        b_off[b] = ok ? (uint32_t)(e * 2) + (uint32_t)(fq << 4) : YOLO_INVALID_OFF;
    }

    const int KT = p.cin_chunks >> 3;           // 64-channel super-steps
    // Every iteration issues the SAME number of requests -- beyond the last super-step with an out-of-range offset (zeros into a free
    // slot / dead registers): with conditional issues the compiler's own waitcnt insertion, which cannot see the counted waits below,
    // takes the path with the fewest younger requests and puts `s_waitcnt vmcnt(0)` right behind the loads just issued.
    auto issue_a = [&](int kt, int slot) {
        const uint32_t ka = (uint32_t)kt * ROWB;
        const bool live = kt < KT;
#pragma unroll
        for (int j = 0; j < JA; ++j) x1_dma16(rs_w, smem + slot * A_BYTES + (j * 8 + wave) * 1024, live ? a_off[j] : YOLO_INVALID_OFF, ka);
    };
    uint4v bq[2][TP][2];                        // [register set = super-step parity][pixel fragment][k-step]
    auto issue_b = [&](int kt, int set, int s) {
        const uint32_t kb = (uint32_t)kt * ROWB + (uint32_t)s * 64u;
        const bool live = kt < KT;
#pragma unroll
        for (int b = 0; b < TP; ++b)
            bq[set][b][s] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rs_in, live ? b_off[b] : YOLO_INVALID_OFF, kb, 0));
    };

    float4v acc[TM][TP];
    conv_init_acc_bias<TM, TP>(p, acc, n0 + fq * CH);
    const int fswz = (fr >> 1) & 7;
    const unsigned char *const a_base = smem + fr * ROWB;

    // one super-step; the register set and the ring slot are compile-time constants
    auto step = [&](int kt, auto setc, auto slotc) {
        constexpr int set = decltype(setc)::value, slot = decltype(slotc)::value;
        // younger than what this super-step needs: exactly the requests of the previous iteration -- weights of kt + 2, pixels of kt + 1
        x1_wait_vm<NI>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();           // stage kt visible to every wave; the slot of stage kt - 1 is no longer read
        issue_a(kt + 3, (slot + 3) % S);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int so = (((s * 4 + fq) ^ fswz) & 7) << 4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {       // the A fragments in two halves of four (16 instead of 32 registers)
                uint4v fa[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const uint4v *>(a_base + slot * A_BYTES + (h * 4 + a) * 16 * ROWB + so);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < TP; ++b) acc[h * 4 + a][b] = mma_chunk<T>(fa[a], bq[set][b][s], acc[h * 4 + a][b]);
            }
            issue_b(kt + 2, set, s);
        }
    };

    // ---- prologue: weights of super-steps 0..2, pixels of 0 and 1 (request order = the loop's: A, B per iteration)
    // (pinned in this order: the compiler's waitcnt insertion merges the prologue's request order with the loop's at the loop header
    // and waits for the smaller count of younger requests -- with the pixel loads sunk behind the weight DMAs here, the first step of
    // every trip through the unrolled loop would wait for part of the NEXT super-step's requests)
    issue_a(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    issue_b(0, 0, 0); issue_b(0, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    issue_a(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    issue_b(1, 1, 0); issue_b(1, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    issue_a(2, 2);
    __builtin_amdgcn_sched_barrier(0);
    // whole trips of four super-steps without an exit in between (an exit edge inside the unrolled body is, to the compiler's waitcnt
    // insertion, a path on which fewer requests were issued: it then waits for requests of the NEXT super-step at three of the four
    // steps), then the 0..3 remaining super-steps as straight-line code
    int kt = 0;
    for (; kt + 4 <= KT; kt += 4) {
        step(kt, std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
        step(kt + 1, std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
        step(kt + 2, std::integral_constant<int, 0>(), std::integral_constant<int, 2>());
        step(kt + 3, std::integral_constant<int, 1>(), std::integral_constant<int, 3>());
    }
    if (kt < KT) step(kt, std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
    if (kt + 1 < KT) step(kt + 1, std::integral_constant<int, 1>(), std::integral_constant<int, 1>());
    if (kt + 2 < KT) step(kt + 2, std::integral_constant<int, 0>(), std::integral_constant<int, 2>());

    if constexpr (EPI == 1) conv_epilogue_fast<TM, TP, 0>(p, acc, n0 + fq * CH, m0 + wave * (TP * 16), fr);
    else conv_epilogue<T, TM, TP, 0, true>(p, acc, n0 + fq * CH, m0 + wave * (TP * 16), fr);
}

bool conv_1x1_regb_ok(const ConvParams &p) {
    return !p.f32 && !p.out_f32 && p.ksize == 1 && p.stride == 1 && p.cin_chunks >= 8 && p.cin_chunks % 8 == 0 && p.ksplit <= 1 &&
           (p.in_ld % 8) == 0 && (p.in_coff % 8) == 0 && p.in_bytes != 0;
}

const char *conv_1x1_regb_symbol(bool fast) {
    return fast ? "void yolo::conv1x1_regb_kernel<1>(yolo::ConvParams)" : "void yolo::conv1x1_regb_kernel<0>(yolo::ConvParams)";
}

hipError_t launch_conv_1x1_regb(const ConvParams &p, hipStream_t s) {
    if (!conv_1x1_regb_ok(p)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)p.n_blocks), block(512);
    if (conv_fast_epilogue_ok(p)) hipLaunchKernelGGL((conv1x1_regb_kernel<1>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv1x1_regb_kernel<0>), grid, block, 0, s, p);
    return hipGetLastError();
}

}  // namespace yolo
