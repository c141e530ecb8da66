// 3x3 / stride-1 implicit-GEMM conv (fp16) with TAP REUSE of the pixel operand.
//
// Why: the LDS-DMA kernels of conv_dma.hip fetch the pixel tile once PER TAP (im2col on the fly), and their K
// loop is bound by the L2 -> LDS path: 20 TB/s chip-wide = ~78 GB/s per CU measured with the MFMAs switched
// off (profiles/r01_ablation.md; the guide's L2-served gather-into-LDS figure is 66-73 GB/s per CU), against
// ~85-130 flop/byte of those tiles.  Here the input patch of a block is loaded ONCE per 32-channel slice and
// all nine taps read their MFMA B operand from it at a row offset, so a 128-cout x 256-pixel block moves
// 8 KiB of weights per tap + ~26 KiB of pixels per nine taps: ~190 flop/byte, 2.2x fewer L2 -> LDS bytes.
//
// Pixel indexing: positions q of a PADDED-LINEAR grid [n][y in 0..H][x in 0..W] -- one shared pad column after
// every image row and one shared pad row after every image.  Tap (kh, kw) of position q is position
// q + (kh-1)(W+1) + (kw-1), for every q, so a block owning NB consecutive positions keeps the NB + 2W + 4
// positions around them in LDS (64-byte rows = 32 channels) and tap t reads rows shifted by kh (W+1) + kw.
// Pad positions are filled with zeros by the LDS-DMA range check (per-lane source offset = invalid) -- this is
// the conv's zero padding -- and as OUTPUT positions they are computed and dropped in the epilogue
// ((H+1)(W+1)/(HW) - 1 wasted MFMA work: 2.6 % at 76x76, 10.8 % at 19x19).
//
// LDS swizzle of the patch: logical 16-byte chunk c of row R sits at chunk c ^ (2 * ((R >> 2) & 1)).  Unlike
// the table swizzle of the weight tile this one stays conflict-free for ds_read_b128 under ANY row shift: the
// hardware services lanes {0-3, 12-15, 20-27} (k-chunk q for fragment rows 0-3, 12-15, chunk q^1 for rows
// 4-11) together, the four rows of one residue class mod 4 are consecutive values of R >> 2, and
// {s(u), s(u+1)^1, s(u+2)^1, s(u+3)} is a permutation of 0..3 for s(u) = 2 (u & 1) and every u.
//
// K loop: for each 32-channel slice c: for tap 0..8 (unrolled): weights of (tap, c) come through a 3-slot
// LDS-DMA ring (slot = tap % 3), the patch of slice c+1 is fetched into the other patch buffer during the taps
// of slice c.  One barrier per tap; vmcnt waits are counted so that the patch may stay in flight for three taps.
#include "conv_tap_tile.h"

namespace yolo {

template <bool F32, int WM, int WN, int TM, int TP, int PRG, int OCC, int MODE, bool SPLITK, bool FAST = false, bool FUSE2 = false>
__global__ void __launch_bounds__(WM * WN * 64, OCC) conv3x3_tap_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[conv3x3_tap_lds_bytes<WM, TM, TP, PRG, FUSE2>()];
    conv3x3_tap_tile<F32, WM, WN, TM, TP, PRG, OCC, MODE, SPLITK, FAST, FUSE2>(p, smem, xcd_remap(blockIdx.x, p.n_blocks), (int)blockIdx.x, (int)blockIdx.y,
                                                                               (int)gridDim.y);
}

// PAIRED form (round 5, profiles/r05_ablation.md section 15).  Two co-resident workgroups of the kernel above do NOT share a CU evenly: the block trace
// split by threadgroup id shows the older one running its K loop at the lone rate (0.52 us per tap) and the younger one at 0.4 of that until the first
// has gone -- the matrix pipes are ~68 % busy inside the K loops.  Here the same two tiles are the two HALVES of one 16-wave workgroup (each half its
// own LDS rings, the tile function unchanged), every barrier meets all sixteen waves, a second barrier sits in the middle of each tap's MFMA burst, and
// half 1 runs one barrier behind half 0: at every barrier one half has eight MFMAs per wave ready to issue while the other starts its fragment reads.
template <int WM, int WN, int TM, int TP, int PRG, int MODE>
__global__ void __launch_bounds__(1024, 4) conv3x3_tap_pair_kernel(const ConvParams p) {
    constexpr int LDS = conv3x3_tap_lds_bytes<WM, TM, TP, PRG, false>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * LDS];
    const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 9));
    const int wg = xcd_remap(blockIdx.x, (p.n_blocks + 1) >> 1);
    int t = 2 * wg + half;
    const bool valid = t < p.n_blocks;          // (an odd number of tiles: the last workgroup's second half runs the last tile again and stores nothing)
    if (!valid) t = p.n_blocks - 1;
    conv3x3_tap_tile<false, WM, WN, TM, TP, PRG, 4, MODE, false, true, false, false, false, true>(p, smem + half * LDS, t, t, 0, 1, half, valid);
}

// ---- persistent ("stream") form ---------------------------------------------------------------------------------------------
// Block traces (tools/trace_blocks.py, profiles/r03_ablation.md) show what a workgroup of the kernel above does outside its K loop:
// ~0.8 us of setup, ~1.5-2 us until its first patch + weights have landed, 3.6-3.9 us of epilogue -- 20 % of its life at 76 x 76,
// half of it at 304 x 304 (K = 288) -- and a CU is only at full rate while BOTH of its workgroups are inside their K loops.
// Here the 512 (or 256) resident workgroups stay and walk tiles blockIdx, blockIdx + grid, ...: the (tile, channel slice) items
// form ONE stream through the same LDS rings, so the patch and the first two weight tiles of the next tile are requested during the
// last slice of the current one and are in LDS when its epilogue ends; the geometry of the next tile is computed under the K loop.
// The epilogue's loads and stores go through buffer instructions that every wave issues unconditionally (invalid lanes carry an
// out-of-range offset), so their number is a compile-time constant and the counted `s_waitcnt vmcnt` of the two taps that follow an
// epilogue can leave the stores in flight.  The bias of every cout lives in LDS (the accumulators restart from it without a VMEM
// load, which would otherwise have to wait for those stores).  fp16, whole K, OUT_NORMAL, 16-byte aligned views only.
template <int WM, int WN, int TM, int TP, int PRG, int OCC, int MODE>
__global__ void __launch_bounds__(512, OCC) conv3x3_tap_stream_kernel(const ConvParams p) {
    typedef _Float16 T;
    constexpr int NW = 8, S = 3, ROWB = 64;
    constexpr int NA = WM * TM * 16;
    constexpr int NB = WN * TP * 16;
    constexpr int JA_TOT = NA / 16;
    constexpr int JA = (JA_TOT + NW - 1) / NW;
    constexpr int PW = 24;
    constexpr int FROW = MODE == 2 ? PW : 16;
    constexpr int JP = (PRG + NW - 1) / NW;
    constexpr int CH = 4 * TM;
    constexpr int EPC = 8;
    constexpr int NQ = CH / EPC;                // 16-byte chunks of a lane's couts
    constexpr int NST = TP * NQ;                // store instructions of one wave's epilogue (always issued)
    constexpr int A_BYTES = NA * ROWB;
    constexpr int P_BYTES = PRG * 1024;
    constexpr int BIAS_FLOATS = kTapStreamBiasFloats;
    static_assert(WM * WN == NW && (JA_TOT % NW == 0 || JA_TOT < NW), "tile / wave mapping");
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * A_BYTES + 2 * P_BYTES + BIAS_FLOATS * 4];
    unsigned char *const smemP = smem + S * A_BYTES;
    float *const sbias = reinterpret_cast<float *>(smem + S * A_BYTES + 2 * P_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool has_a = JA_TOT % NW == 0 || wave < JA_TOT;   // wave-uniform
    const int fr = lane & 15, fq = lane >> 4;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.res ? p.res : p.out), 0, p.res ? p.res_bytes : 0u, 0x00020000);

    for (int i = tid; i < p.cout_pad && i < BIAS_FLOATS; i += 512) sbias[i] = p.bias[i];      // (read after the first tap's barrier at the earliest)

    // ---- lane constants of the DMA geometry --------------------------------------------------------
    const int lrow = lane >> 2;
    uint32_t a_off[JA];         // without the cout tile: n0 * wrow_bytes rides in the scalar offset of the DMA
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int r = (j * NW + wave) * 16 + lrow;
        const int ws = r / (TM * 16), R = r % (TM * 16);
        const int tm = R >> 4, g4 = (R >> 2) & 3, jj = R & 3;
        const int ch = ws * (TM * 16) + g4 * CH + 4 * tm + jj;
        a_off[j] = (uint32_t)ch * p.wrow_bytes + (uint32_t)(((lane & 3) ^ tap_swz_w(lrow)) << 4);
    }
    constexpr int JP_FULL = PRG - (JP - 1) * NW;
    const bool jp_full = wave < JP_FULL;
    const uint32_t csw_p = (uint32_t)(((lane & 3) ^ (((lrow >> 2) & 1) << 1)) << 4);
    uint32_t b_off[JP];
    // per-tile: patch offsets of tile `bid` (remapped linear tile id) -> b_off; returns q0 and n0 of the tile
    auto geometry = [&](int bid, int &q0, int &n0) __attribute__((always_inline)) {
        const int mt = (int)fdiv((uint32_t)bid, p.dtiles_n);
        n0 = (bid - mt * p.n_tiles_n) * NA;
        q0 = mt * NB;
        int t2_n = 0, t2_y0 = 0, t2_x0 = 0;
        if (MODE == 2) {
            t2_n = (int)fdiv((uint32_t)mt, p.dqHW);
            const int r = mt - t2_n * p.qHW;
            const int ty = (int)fdiv((uint32_t)r, p.dqW);
            t2_y0 = ty * (NB / 16);
            t2_x0 = (r - ty * p.qW) * 16;
        }
#pragma unroll
        for (int j = 0; j < JP; ++j) {
            const int g = j * NW + wave;
            bool ok;
            int n, y, x;
            if (MODE == 2) {
                const int R = g * 16 + lrow;
                const int pr = R / PW, pc = R - pr * PW;
                n = t2_n; y = t2_y0 - 1 + pr; x = t2_x0 - 1 + pc;
                ok = g < PRG && pc < 18 && pr < NB / 16 + 2 && t2_n * p.HoWo < p.M && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            } else {
                const int q = q0 - (p.qW + 1) + g * 16 + lrow;
                ok = g < PRG && q >= 0 && q < p.Mq;
                const int qq = ok ? q : 0;
                n = (int)fdiv((uint32_t)qq, p.dqHW);
                const int r = qq - n * p.qHW;
                y = (int)fdiv((uint32_t)r, p.dqW);
                x = r - y * p.qW;
                ok = ok && x < p.W && y < p.H;
            }
            const long long e = (long long)n * p.in_img_stride + ((long long)y * p.W + x) * p.in_ld + p.in_coff;
            b_off[j] = ok ? (uint32_t)(e * 2) + csw_p : YOLO_INVALID_OFF;
        }
    };

    const int C = p.cin_chunks >> 2;            // channel slices per tile
    auto issue_patch = [&](int c, int buf) __attribute__((always_inline)) {
        const uint32_t koff = (uint32_t)c * ROWB;
#pragma unroll
        for (int j = 0; j < JP; ++j)
            if (j + 1 < JP || jp_full) tap_dma16(rs_in, smemP + buf * P_BYTES + (j * NW + wave) * 1024, b_off[j], koff);
    };
    auto issue_weights = [&](int tap, int c, int n0, int slot) __attribute__((always_inline)) {
        const uint32_t ka = (uint32_t)(tap * p.cin_chunks + 4 * c) * 16 + (uint32_t)n0 * p.wrow_bytes;
        if (has_a) {
#pragma unroll
            for (int j = 0; j < JA; ++j) tap_dma16(rs_w, smem + slot * A_BYTES + (j * NW + wave) * 1024, a_off[j], ka);
        }
    };

    float4v acc[TM][TP];
    const int a_frag = (wm * TM * 16 + fr) * ROWB + (((fq ^ tap_swz_w(fr)) & 3) << 4);
    const int rb = wn * TP * FROW + fr;
    const int c_lane = wm * (TM * 16) + fq * CH;        // first cout of this lane inside the cout tile
    auto compute = [&](int slot, int buf, int shift) __attribute__((always_inline)) {
        const unsigned char *A = smem + slot * A_BYTES + a_frag;
        const int R = rb + shift;
        const unsigned char *B = smemP + buf * P_BYTES + (R << 6) + ((fq << 4) ^ ((R & 4) << 3));
        uint4v fa[TM], fb[TP];
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const uint4v *>(A + a * 16 * ROWB);
#pragma unroll
        for (int b = 0; b < TP; ++b) fb[b] = *reinterpret_cast<const uint4v *>(B + b * FROW * ROWB);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b) tap_mfma<T>(acc[a][b], fa[a], fb[b]);
    };
    // epilogue of one tile: leaky, + residual, fp16, 16-byte stores; NQ * TP loads (if any) and NST stores per wave, always
    auto epilogue = [&](int q0, int n0) __attribute__((always_inline)) {
        // the residual of DEPTH fragments is in flight at a time: all of them for the TP = 2 tile; two for the TP = 4 tiles, whose 128
        // registers cannot hold 32 residual registers next to the K loop's state that has to survive the epilogue here
        constexpr int DEPTH = TP > 2 ? 2 : TP;
        const int cbase = n0 + c_lane;
        const bool c_ok = cbase < p.Cout;
        uint32_t ooff[DEPTH];
        uint4v rv[DEPTH][NQ];
        auto request = [&](int b, int slot) __attribute__((always_inline)) {
            int n, rem, oy, ox;
            const bool ok = conv_decode_pixel<MODE>(p, q0 + wn * (TP * 16) + b * 16 + fr, n, rem, oy, ox) && c_ok;
            const long long o = ((long long)n * p.out_img_stride + (long long)rem * p.out_ld + cbase) * 2;
            ooff[slot] = ok ? (uint32_t)o : YOLO_INVALID_OFF;
            if (p.has_res) {
                const long long ro = ((long long)n * p.res_img_stride + (long long)rem * p.res_ld + cbase) * 2;
                const uint32_t roff = ok ? (uint32_t)ro : YOLO_INVALID_OFF;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    rv[slot][q] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rs_res, roff, q * 16, 0));
            }
        };
#pragma unroll
        for (int b = 0; b < DEPTH; ++b) request(b, b);
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            const int slot = b % DEPTH;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                T t[EPC], r[EPC];
                if (p.has_res) __builtin_memcpy(r, &rv[slot][q], 16);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const int i = q * EPC + e;
                    float x = acc[i >> 2][b][i & 3];
                    x = p.leaky ? fmaxf(0.1f * x, x) : x;
                    if (p.has_res) x += (float)r[e];
                    t[e] = (T)x;
                }
                uint4v u;
                __builtin_memcpy(&u, t, 16);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, u), rs_out, ooff[slot], q * 16, 0);
            }
            if (b + DEPTH < TP) {
                __builtin_amdgcn_sched_barrier(0);      // (keeps the next request behind this fragment's use of the slot)
                request(b + DEPTH, slot);
            }
        }
    };
    auto init_acc = [&](int n0, bool from_lds) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const float4v bv = from_lds ? *reinterpret_cast<const float4v *>(sbias + n0 + c_lane + 4 * a)
                                        : *reinterpret_cast<const float4v *>(p.bias + n0 + c_lane + 4 * a);
#pragma unroll
            for (int b = 0; b < TP; ++b) acc[a][b] = bv;
        }
    };

    // ---- the stream ------------------------------------------------------------------------------------
    int it = blockIdx.x;                       // this workgroup's tiles: it, it + grid, ... (< n_blocks)
    int q0_cur, n0_cur, q0_nxt = 0, n0_nxt = 0;
    geometry(xcd_remap(it, p.n_blocks), q0_cur, n0_cur);
    init_acc(n0_cur, false);
    issue_patch(0, 0);
    issue_weights(0, 0, n0_cur, 0);
    issue_weights(1, 0, n0_cur, 1);
    int c = 0;
    bool after_epi = false;                    // the previous item ended with an epilogue: NST stores sit between its DMAs and ours
    bool done = false;

    // one (tile, slice) item; the patch buffer index is a compile-time constant
    auto run_item = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const bool last_c = c + 1 == C;
        const int it_n = last_c ? it + (int)gridDim.x : it;
        const bool more = !last_c || it_n < p.n_blocks;
        const int c_n = last_c ? 0 : c + 1;
        if (!last_c) { n0_nxt = n0_cur; q0_nxt = q0_cur; }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bool last = !more && tap == 8;
            const bool with_patch = more && (tap == 1 || tap == 2);
            const bool st = after_epi && tap < 2;      // the epilogue's stores are younger than the DMAs this tap waits for
            if (last) tap_wait_vm<0>();
            else if (has_a) {
                if (!with_patch) { if (st) tap_wait_vm<JA + NST>(); else tap_wait_vm<JA>(); }
                else if (jp_full) { if (st) tap_wait_vm<JA + JP + NST>(); else tap_wait_vm<JA + JP>(); }
                else { if (st) tap_wait_vm<JA + JP - 1 + NST>(); else tap_wait_vm<JA + JP - 1>(); }
            } else {
                if (!with_patch) { if (st) tap_wait_vm<NST>(); else tap_wait_vm<0>(); }
                else if (jp_full) { if (st) tap_wait_vm<JP + NST>(); else tap_wait_vm<JP>(); }
                else { if (st) tap_wait_vm<JP - 1 + NST>(); else tap_wait_vm<JP - 1>(); }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (tap == 0 && more) {
                if (last_c) geometry(xcd_remap(it_n, p.n_blocks), q0_nxt, n0_nxt);      // b_off of the finished tile is dead: its last patch is in LDS
            }
            {   // weights two taps ahead (the stream continues into the next item)
                if (tap + 2 < 9) issue_weights(tap + 2, c, n0_cur, (tap + 2) % S);
                else if (more) issue_weights(tap + 2 - 9, c_n, n0_nxt, (tap + 2) % S);
            }
            if (tap == 0 && more) issue_patch(c_n, buf ^ 1);
            const int kh = tap / 3, kw = tap - 3 * kh;
            compute(tap % S, buf, MODE == 2 ? kh * PW + kw : kh * p.qW + kw);
        }
        after_epi = false;
        if (last_c) {
            tap_mfma_drain<TM, TP>(acc);
            epilogue(q0_cur, n0_cur);
            if (!more) { done = true; return; }
            it = it_n; q0_cur = q0_nxt; n0_cur = n0_nxt;
            init_acc(n0_cur, true);
            after_epi = true;
        }
        c = c_n;
    };
    for (;;) {
        run_item(std::integral_constant<int, 0>());
        if (done) break;
        run_item(std::integral_constant<int, 1>());
        if (done) break;
    }
}

// (6 = 256 couts x 224 positions, one workgroup per CU: 19 x 19 maps at batch 32 are 12 800 padded positions -> 58 x 4 = 232
// tiles on 256 CUs, where the 256-position tiles leave 200 or 400 workgroups on 256 / 512 slots)
// variants: 0 = 128 couts x 256 positions, 1 = 256 x 256 (one workgroup per CU), 2 = 128 x 192, 3 = 128 x 128 (smaller
// position tiles fill the 512 workgroup slots of the chip better on small feature maps), all padded-linear;
// 4 = 128 x (16 x 16) and 5 = 64 x (16 x 16) 2-D tiles for maps wider than 78 (any width)
// 7 = 128 couts x (8 x 16) 2-D tile at THREE workgroups per CU (48 KiB LDS, <= 80 VGPRs) for wide maps with a short K, where
// a workgroup spends as long in setup + epilogue as in its K loop (152 x 152 64 -> 128: block trace in profiles/r03_ablation.md)
// 8 = 32 couts x (16 x 16): the one 3x3 layer with 32 filters behind the first conv (tiny-YOLOv2 16 -> 32 at 208 x 208; float32 MFMA is 1/16 of
// fp16's, so the 64-cout tile's idle half would double a launch that is MFMA-bound)
// 9 = 128 couts x 384 positions, IMAGE-ALIGNED, one workgroup per CU: a tile is one whole image of a map with H (W+1) <= 384 (19 x 19:
// 380), tile m starts at position m (H+1)(W+1) -- the shared pad row behind every image is never computed (6 % of the positions are
// padding instead of 10.8 %) and 19 x 19 at batch 32 is 32 x 8 = 256 tiles: every CU busy, 14 % less work per CU than the 232 tiles
// of variant 6
// 10 = 128 x 256 and 11 = 128 x 384 image-aligned for 3x3 / STRIDE 2 (MODE 4: parity planes of the input, see the kernel): fp16 only
// 12 = 128 x 192 image-aligned: one 12 x 12 or 13 x 13 image per tile (YOLOv2-416 / YOLOv3-416 tails: 13 x 14 = 182 of 192 positions
// real, where 256-position tiles of the padded-linear grid compute 23 % padding); with the in-launch pair split 16 images x 8 cout
// tiles x 2 K halves = 256 workgroups
// 13 = variant 10 with a patch of 26 row groups (output maps up to 158 wide): the stride-2 conv into the 152 x 152 stage, whose
// workgroups hold all 128 couts of 256 positions -- the one stride-2 tile with the back-to-back 1x1 instantiation (FUSE2)
// (round 5, measured and NOT kept -- profiles/r05_ablation.md, source in commit 57e38ee: variant 0's block tile as four fat waves of 64 couts x 128
// positions with the position fragments prefetched across the barrier, +0.5 ... +2.5 % slower; variant 9's 128 x 384 tile on the padded-linear grid
// for 38 x 38 maps -- 508 tiles on 256 CUs -- +4.3 % slower than the 764 workgroups of variant 0; position fragments of the taps kw = 1, 2 by DPP lane
// shifts instead of LDS reads, correct and +5 ... +9 % slower)
static const int kTapNB[] = {256, 256, 192, 128, 256, 256, 224, 128, 256, 384, 256, 384, 192, 256};
static const int kTapPRG[] = {26, 26, 26, 28, 27, 27, 17, 12, 27, 27, 21, 26, 14, 26};
static const int kTapTP[] = {4, 4, 3, 2, 4, 2, 7, 2, 2, 6, 4, 6, 3, 4};        // TP of the variant (YOLO_TAP_VARIANTS below): position fragments per wave
static const int kTapVariants = 14;
static const bool kTapF32[] = {false, false, false, true, false, true, false, false, true, false, false, false, false, false};      // float32 tiles: TP <= 2 (second-level accumulator)
bool conv_tap_image_aligned(int variant) { return variant == 9 || variant == 11 || variant == 12; }
bool conv_tap_stride2(int variant) { return variant == 10 || variant == 11 || variant == 13; }
bool conv_tap_splitk_ok(int variant) { return variant == 3; }      // the 128 x 128 tile has the (two-pass) split-K instantiation
bool conv_tap_pair_ok(int variant, bool f32) { return variant == 3 || ((variant == 0 || variant == 12) && !f32); }   // in-launch pair split: also the fp16 128 x 256 and image-aligned 128 x 192 tiles
bool conv_tap_is2d(int variant) { return variant == 4 || variant == 5 || variant == 7 || variant == 8; }
bool conv_tap_f32_ok(int variant) { return variant >= 0 && variant < kTapVariants && kTapF32[variant]; }
bool conv_tap_fits(int variant, int W) {
    if (variant < 0 || variant >= kTapVariants) return false;
    if (conv_tap_is2d(variant)) return true;
    if (conv_tap_stride2(variant)) {        // W = the INPUT's width (even); the position grid is the output's
        if (W & 1) return false;
        W >>= 1;
    }
    // (the image-aligned tile: square maps of 17 .. 19 -- a whole image per tile with at least 80 % of the positions real)
    if (conv_tap_image_aligned(variant) && (W * (W + 1) > kTapNB[variant] || W * (W + 1) * 5 < kTapNB[variant] * 4)) return false;
    if (conv_tap_stride2(variant)) return kTapNB[variant] + W + 2 <= kTapPRG[variant] * 16;
    // (position-interleaved fragments: the patch buffer holds TP planes of PRG * 16 / TP whole rows)
    const int tp = kTapTP[variant], rows = YOLO_TAP_PIL ? kTapPRG[variant] * 16 / tp * tp : kTapPRG[variant] * 16;
    return kTapNB[variant] + 2 * W + 4 <= rows;
}

// variant id, then the template arguments after F32: WM, WN, TM, TP, PRG, OCC, MODE (written with ", " so that the
// stringified list equals the demangled symbol)
#define YOLO_TAP_VARIANTS(X) \
    X(0, 2, 4, 4, 4, 26, 4, 1) \
    X(1, 2, 4, 8, 4, 26, 2, 1) \
    X(2, 2, 4, 4, 3, 26, 4, 1) \
    X(3, 2, 4, 4, 2, 28, 4, 1) \
    X(4, 2, 4, 4, 4, 27, 4, 2) \
    X(5, 1, 8, 4, 2, 27, 4, 2) \
    X(6, 4, 2, 4, 7, 17, 2, 1) \
    X(7, 2, 4, 4, 2, 12, 6, 2) \
    X(8, 1, 8, 2, 2, 27, 4, 2) \
    X(9, 2, 4, 4, 6, 27, 2, 1) \
    X(10, 2, 4, 4, 4, 21, 4, 4) \
    X(11, 2, 4, 4, 6, 26, 2, 4) \
    X(12, 2, 4, 4, 3, 14, 4, 1) \
    X(13, 2, 4, 4, 4, 26, 4, 4)

const char *conv_tap_symbol(int variant, bool f32, bool fast) {
    switch (variant) {
#define X(id, ...) case id: return f32 ? "void yolo::conv3x3_tap_kernel<true, " #__VA_ARGS__ ", false, false, false>(yolo::ConvParams)" \
                                       : fast ? "void yolo::conv3x3_tap_kernel<false, " #__VA_ARGS__ ", false, true, false>(yolo::ConvParams)" \
                                              : "void yolo::conv3x3_tap_kernel<false, " #__VA_ARGS__ ", false, false, false>(yolo::ConvParams)";
        YOLO_TAP_VARIANTS(X)
#undef X
    default: return "";
    }
}

// the persistent form (conv3x3_tap_stream_kernel) is instantiated for variant 5 only (64 couts x 16 x 16 pixels, 93 VGPRs, no spill:
// 304 x 304 32 -> 64 at batch 32 0.242 -> 0.225 ms).  The TP = 4 tiles (variants 0 and 4) sit at the 128-register limit of two
// workgroups per CU: with the loop state of the stream they spill (19-23 VGPRs with all four residual fragments in flight, 5-17 with
// the two-deep residual pipeline the epilogue has for them), reloads land inside the K loop (each a `s_waitcnt vmcnt(0)` that drains
// the DMA pipeline) and the launches got SLOWER both times: 76 x 76 +15 % / +3 %, 152 x 152 +6.5 % / +13 % (profiles/r03_ablation.md).
// (round 5: with the MFMAs in place the 128 x 256 tile compiles in this form without a spill -- 123 registers -- and is still slower than the
// plain kernel: 76 x 76 +3 ... +5 %, 38 x 38 +-0, profiles/r05_ablation.md section 6; so it stays the 64-cout tile's alone)
#define YOLO_TAP_STREAM_VARIANTS(X) \
    X(5, 1, 8, 4, 2, 27, 4, 2)

const char *conv_tap_stream_symbol(int variant) {
    switch (variant) {
#define X(id, ...) case id: return "void yolo::conv3x3_tap_stream_kernel<" #__VA_ARGS__ ">(yolo::ConvParams)";
        YOLO_TAP_STREAM_VARIANTS(X)
#undef X
    default: return "";
    }
}

bool conv_tap_stream_ok(const ConvParams &p, int variant) {
    if (variant != 5) return false;
    if (p.f32 || p.out_f32 || p.ksplit > 1 || p.outmode != OUT_NORMAL || !p.vec_out || (p.has_res && (!p.vec_res || !p.res_bytes)) || !p.out_bytes) return false;
    if (p.Cout % 16 || (p.Cout + 127) / 128 * 128 > kTapStreamBiasFloats) return false;
    return true;
}

static hipError_t launch_conv_tap_stream(const ConvParams &p0, int variant, hipStream_t s) {
    ConvParams p = p0;
    p.cout_pad = (p.Cout + 127) / 128 * 128;
    const int slots = 512;              // two workgroups per CU (every stream variant is OCC 4)
    const dim3 grid((unsigned)(p.n_blocks < slots ? p.n_blocks : slots));
    switch (variant) {
#define X(id, ...) case id: hipLaunchKernelGGL((conv3x3_tap_stream_kernel<__VA_ARGS__>), grid, dim3(512), 0, s, p); break;
        YOLO_TAP_STREAM_VARIANTS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_tap(const ConvParams &p0, int variant, hipStream_t s) {
    ConvParams p = p0;
    static const bool no_fast_epi = getenv("YOLO_NO_FAST_EPI") != nullptr;        // A/B switch, read once (same results either way)
    // (variant 10, the two-per-CU stride-2 tile: its lean instantiation spills 13 registers at the 128-register limit; the generic one does not)
    p.fast_epi = (!no_fast_epi || p.fuse2) && conv_fast_epilogue_ok(p) && variant != 10 && (variant != 13 || p.fuse2) ? 1 : 0;      // (the fused pair exists in the lean form only)
    const bool s2 = conv_tap_stride2(variant);
    if (p.ksize != 3 || p.stride != (s2 ? 2 : 1) || p.pad != 1 || (p.cin_chunks & 3) || !conv_tap_fits(variant, p.W) || (p.f32 && !conv_tap_f32_ok(variant)))
        return hipErrorInvalidValue;
    if (s2 ? ((p.H & 1) || (p.W & 1) || p.Ho * 2 != p.H || p.Wo * 2 != p.W || p.f32 || p.ksplit > 1 || (p.fuse2 && variant != 13) || p.outmode == OUT_POOL2 || p.qW != p.Wo + 1)
           : (p.Ho != p.H || p.Wo != p.W))
        return hipErrorInvalidValue;
    if (conv_tap_image_aligned(variant) ? (p.q_stride != p.qHW || p.Ho * (p.Wo + 1) > kTapNB[variant]) : (!conv_tap_is2d(variant) && p.q_stride != kTapNB[variant]))
        return hipErrorInvalidValue;
    if (p.outmode == OUT_POOL2 && ((variant != 4 && variant != 5 && variant != 8) || (p.H & 1) || (p.W & 1) || p.has_res || p.ksplit > 1 || !p.vec_out || p.Cout % 16))
        return hipErrorInvalidValue;        // the fused pool lives in the 2-D tiles' epilogue only (plan.cpp asks for it accordingly)
    if (p.stream && conv_tap_stream_ok(p, variant)) return launch_conv_tap_stream(p, variant, s);
    {
        static const bool pair_on = getenv("YOLO_TAP_PAIRED") != nullptr;       // experiment switch
        if (pair_on && variant == 0 && !p.f32 && p.fast_epi && p.ksplit <= 1 && !p.fuse2 && p.outmode == OUT_NORMAL) {
            hipLaunchKernelGGL((conv3x3_tap_pair_kernel<2, 4, 4, 4, 26, 1>), dim3((unsigned)((p.n_blocks + 1) / 2)), dim3(1024), 0, s, p);
            return hipGetLastError();
        }
    }
    const dim3 grid((unsigned)p.n_blocks, (unsigned)(p.ksplit > 1 ? p.ksplit : 1));
    if (p.ksplit > 1) {     // split-K instantiation (128 x 128 tile)
        if (!(p.pair ? conv_tap_pair_ok(variant, p.f32 != 0) : conv_tap_splitk_ok(variant)) || !p.part || p.kunits < 1 ||
            (long long)p.ksplit * p.kunits < (p.cin_chunks >> 2))
            return hipErrorInvalidValue;
        if (p.pair && (p.ksplit < 2 || (p.ksplit > 2 && variant != 3) || !p.pair_cnt || p.n_blocks > 512 ||        // (512 padded tickets: api.cpp kPairCounterBytes)
                       (unsigned long long)p.n_blocks * (unsigned long long)p.ksplit * 128ull * kTapNB[variant] * 4ull > p.part_bytes))
            return hipErrorInvalidValue;
        // (OCC 2 = up to 256 registers: the pair launches are <= 512 workgroups of half K on 256 CUs, and the 128 x 256 tile + the
        // hand-off state spills at the 128 registers of two-per-CU residency)
        if (variant == 0) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 4, 26, 2, 1, true>), grid, dim3(512), 0, s, p);
        else if (variant == 12) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 3, 14, 2, 1, true>), grid, dim3(512), 0, s, p);
        else if (p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<true, 2, 4, 4, 2, 28, 4, 1, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 2, 28, 4, 1, true>), grid, dim3(512), 0, s, p);
        return hipGetLastError();
    }
    if (p.outmode == OUT_POOL2) {       // MODE 3 = the 2-D tile + the max-pool in the epilogue: variants 4 (fp16) and 5 (fp16, float32)
        if (variant == 4 && !p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 4, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else if (variant == 5 && !p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 1, 8, 4, 2, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else if (variant == 5) hipLaunchKernelGGL((conv3x3_tap_kernel<true, 1, 8, 4, 2, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else if (variant == 8 && !p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 1, 8, 2, 2, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else if (variant == 8) hipLaunchKernelGGL((conv3x3_tap_kernel<true, 1, 8, 2, 2, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (p.fuse2) {          // back-to-back 1x1: the 2-D 128 x 256 tile (residual block's 3x3) or the wide stride-2 tile (no residual), lean epilogue
        if ((variant != 4 && variant != 13) || !p.fast_epi || (variant == 4) != (p.has_res != 0) || p.n_tiles_n != 1 || p.Cout != 128 || !p.w2 || !p.b2 || !p.out2 || !p.out2_bytes)
            return hipErrorInvalidValue;
        if (variant == 4) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 4, 27, 4, 2, false, true, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 4, 26, 4, 4, false, true, true>), grid, dim3(512), 0, s, p);
        return hipGetLastError();
    }
    switch (variant) {
#define X(id, ...) case id: \
        if (p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<true, __VA_ARGS__, false>), grid, dim3(512), 0, s, p); \
        else if (p.fast_epi) hipLaunchKernelGGL((conv3x3_tap_kernel<false, __VA_ARGS__, false, true>), grid, dim3(512), 0, s, p); \
        else hipLaunchKernelGGL((conv3x3_tap_kernel<false, __VA_ARGS__, false>), grid, dim3(512), 0, s, p); \
        break;
        YOLO_TAP_VARIANTS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace yolo
