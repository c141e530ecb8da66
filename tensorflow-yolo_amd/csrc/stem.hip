// Darknet-53 stem, fused (fp16 nets): conv 3x3/1 3->32 + BN + leaky on the caller's float32 image, then conv 3x3/2
// 32->64 + BN + leaky (net/v3.py: the first two `conv2d_bn_act` layers; net/layers.py:17-66) in ONE kernel.
//
// Why: unfused, the 32-channel full-resolution tensor (757 MB at 608x608, batch 32) is written by the first kernel
// and read back by the second -- 1.5 GB of the 1.9 GB the two layers move; both ran at ~55 % of the HBM roofline
// (0.34 + 0.38 ms of a 6.2 ms step).  Fused, the tensor only ever exists in LDS: the pair reads 142 MB and writes 378 MB.
//
// Persistent workgroups (8 waves, two per CU) keep the layer-2 weights (9 taps x 64 couts x 64 B, swizzled rows) in
// LDS and walk tiles of 8 x 16 layer-2 outputs x all 64 couts; per tile:
//   phase 0  the 19 x 35 x 3 float32 input patch, fetched into registers one tile ahead, -> fp16 in LDS as 8-byte
//            pixels R G B 0 (zero outside the image = layer 1's padding);
//   phase 1  layer 1 on MFMA for the 17 x 33 positions layer 2 needs: K = 27 (kh, kw, c) spread over two 32-deep
//            k-steps so that a lane's 8 k are two WHOLE neighbouring pixels of one patch row = 16 contiguous bytes
//            (two ds_read_b64; the 2-byte gather of round 1/2 cost 8 conflicting ds_read_u16 + packing per lane:
//            SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.32), bias as the MFMA's C input, leaky, ZERO outside the image (= layer 2's
//            padding), fp16, written as 64-byte rows of an LDS patch whose columns are de-interleaved by parity (all
//            even-column positions, then all odd-column ones), so that the stride-2 taps of 16 consecutive outputs read
//            16 consecutive rows; a group of 16 layer-1 positions has ONE column parity, so it also WRITES 16 consecutive
//            rows (round 3: with the row-pair swizzle both directions are bank-conflict-free; rounds 1-2 wrote rows
//            alternating between the two parity blocks, 3.5 LDS cycles per 8-lane store group instead of 1);
//   phase 2  layer 2: 9 taps x one k-step, A = weight tile of the tap, B = patch rows (shift-invariant swizzle of
//            conv_tap.hip), bias + leaky, 32-byte NHWC stores (a lane owns 16 contiguous couts);
//   phase 3  (when the planner hands it over) the 1x1 64->32 conv + BN + leaky that follows in Darknet-53, straight
//            from the lane's own fp16 outputs as MFMA B operand (no LDS: the k order of an MFMA is free), saving the
//            1x1 kernel its 378 MB re-read of the tensor just written.
#include "conv_common.h"

namespace yolo {

namespace {

typedef __attribute__((address_space(3))) void stem_lds_void;

__device__ __forceinline__ void stem_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char *lds_dst, uint32_t voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (stem_lds_void *)lds_dst, 16, voff, 0, 0, 0);
#else
    (void)rsrc; (void)lds_dst; (void)voff;
#endif
}

__device__ __forceinline__ int stem_swz_w(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }   // weight rows: {0,2,3,1}[(r>>2)&3]
// patch rows: row mod 8 -> (row & 1, chunk) is a bijection, so the eight consecutive rows one ds_write_b128 lane group stores cover all 32
// banks once, and ds_read_b128 of 16 consecutive rows stays conflict-free under ANY row shift (the four rows of a residue class mod 4 in
// one hardware lane group are fq = 0, 1, 1, 0 with (row >> 2) = u .. u + 3: chunks {b, 3 ^ b, 1 ^ b, 2 ^ b} for b = (row >> 1) & 1)
__device__ __forceinline__ int stem_swz_p(int r) { return (r >> 1) & 3; }

constexpr int TY = 8, TX = 16;              // layer-2 outputs per workgroup
constexpr int P1Y = 2 * TY + 1;             // 17 layer-1 rows
constexpr int P1X = 2 * TX + 1;             // 33 layer-1 columns
constexpr int EVEN_COLS = (P1X + 1) / 2;    // 17 even columns per layer-1 row
constexpr int ODD_COLS = P1X / 2;           // 16 odd ones
constexpr int NEVEN = P1Y * EVEN_COLS;      // 289 even-column positions: patch rows 0..288, [py][i] with pitch 17
constexpr int GEVEN = (NEVEN + 15) / 16;    // 19 groups of 16 of them
constexpr int ODD0 = GEVEN * 16;            // odd-column positions: patch rows 304.., [py][i] with pitch 16 (one group = one layer-1 row)
constexpr int NGRP = GEVEN + P1Y;           // 36 groups of 16 positions
constexpr int INY = P1Y + 2, INX = P1X + 2; // 19 x 35 input pixels
constexpr int IN_PX = 36;                   // pixels per input patch row (35 used + one that only zero weights meet)
constexpr int IN_SKEW_PX = 32;              // pixels >= 32 sit 8 bytes further: a lane group's 17 pixel pairs then never share a bank
constexpr int IN_LD = IN_PX * 4 + 4;        // halfs per input patch row: a pixel is R G B 0 = 8 bytes (+ the skew slot)
constexpr int W2_BYTES = 9 * 64 * 64;
constexpr int P_BYTES = NGRP * 16 * 64;
constexpr int IN_BYTES = (INY * IN_LD * 2 + 15) / 16 * 16;     // (the bias table behind it is read 16 bytes at a time)
static_assert((W2_BYTES + P_BYTES + IN_BYTES) % 16 == 0 && (W2_BYTES + P_BYTES) % 16 == 0 && (IN_LD * 2) % 8 == 0,
              "LDS regions read with ds_read_b128 / b64 keep their alignment (a misaligned bias table cost 27 % of the kernel: SQ_LDS_UNALIGNED_STALL)");
static_assert(ODD0 + P1Y * ODD_COLS <= NGRP * 16 && 2 * (W2_BYTES + P_BYTES + IN_BYTES + (64 + 32) * 4) <= 160 * 1024, "patch rows / two workgroups per CU");
constexpr int BIAS_BYTES = (64 + 32) * 4;  // layer-2 and layer-3 biases (read per tile: keeps 24 VGPRs free)

}  // namespace

__global__ void __launch_bounds__(512, 4) stem_v3_kernel(const StemParams p) {
    typedef _Float16 T;
    __shared__ __attribute__((aligned(16))) unsigned char smem[W2_BYTES + P_BYTES + IN_BYTES + BIAS_BYTES];
    unsigned char *const sW = smem;
    unsigned char *const sP = smem + W2_BYTES;
    T *const sIn = reinterpret_cast<T *>(smem + W2_BYTES + P_BYTES);
    float *const sBias = reinterpret_cast<float *>(smem + W2_BYTES + P_BYTES + IN_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // ---- once per (persistent) workgroup -----------------------------------------------------------
    // layer-2 weights -> LDS, [tap][64 rows][64 B]; row r of a tap holds cout 16*((r>>2)&3) + 4*(r>>4) + (r&3)
    {
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2), 0, p.w2_bytes, 0x00020000);
        const int lrow = lane >> 2;
        for (int i = wave; i < 36; i += 8) {
            const int tap = i >> 2, rg = i & 3;
            const int r = rg * 16 + lrow;
            const int co = 16 * ((r >> 2) & 3) + 4 * (r >> 4) + (r & 3);
            const uint32_t off = (uint32_t)co * p.wrow2 + (uint32_t)tap * 64 + (uint32_t)(((lane & 3) ^ stem_swz_w(lrow)) << 4);
            stem_dma16(rs_w, sW + i * 1024, off);
        }
    }
    // layer-1 weights as MFMA A fragments: tile t, row rho = fr holds channel 8*(rho>>2) + 4t + (rho&3).  K layout of
    // k-step s: lane group fq, element j -> patch row kh = (s == 0 ? fq >> 1 : 2), pixel pw = 2 (fq & 1) + (j >> 2),
    // channel c = j & 3; weight zero where pw == 3, c == 3, or (s == 1 and fq >= 2)
    uint4v a1[2][2];
    float4v bias1[2];
    {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ch = 8 * (fr >> 2) + 4 * t + (fr & 3);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                T h[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kh = st == 0 ? (fq >> 1) : 2, pw = 2 * (fq & 1) + (j >> 2), c = j & 3;
                    const bool live = pw < 3 && c < 3 && (st == 0 || fq < 2);
                    const float w = p.w1[(live ? (kh * 3 + pw) * 3 + c : 0) * 32 + ch];      // unconditional load, then select
                    h[j] = live ? (T)w : (T)0.f;
                }
                __builtin_memcpy(&a1[t][st], h, 16);
            }
        }
        bias1[0] = *reinterpret_cast<const float4v *>(p.b1 + 8 * fq);          // channels 8 fq + {0..3}: C input of tile 0
        bias1[1] = *reinterpret_cast<const float4v *>(p.b1 + 8 * fq + 4);
        if (tid < 64) sBias[tid] = p.b2[tid];
        else if (tid < 96 && p.w3) sBias[tid] = p.b3[tid - 64];
        // the fourth half of every pixel, the 36th pixel and the skew slot of every row are never written again: zero (finite) for good
        for (int i = tid; i < INY * (IN_LD / 4); i += 512) *reinterpret_cast<unsigned long long *>(sIn + i * 4) = 0ull;
    }
    // per-lane patch rows of the two B fragments: step 0 row fq >> 1, step 1 row 2 (lane groups 2, 3 meet zero weights there and
    // re-read row 1: finite values); the lane's two pixels are px + 2 (fq & 1) and the next one
    const int brow0 = (fq >> 1) * IN_LD;
    const int brow1 = (fq < 2 ? 2 : (fq >> 1)) * IN_LD;
    auto in_pix = [](int px) { return px * 4 + (px >= IN_SKEW_PX ? 4 : 0); };      // halfs
    const int a_frag = fr * 64 + (((fq ^ stem_swz_w(fr)) & 3) << 4);
    // optional layer 3 (1x1 64->32): the lane already owns 16 channels of its pixel after layer 2, and an MFMA sums over
    // k in any order, so k-step ks takes channels 16 fq + 8 ks + j straight from the lane's registers (no LDS round
    // trip); weight tile t, row rho = fr holds cout 8*(rho>>2) + 4t + (rho&3) -> the lane ends up with 8 contiguous couts
    uint4v a3[2][2];
    if (p.w3) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int co = 8 * (fr >> 2) + 4 * t + (fr & 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                a3[t][ks] = *reinterpret_cast<const uint4v *>(reinterpret_cast<const unsigned char *>(p.w3) + co * 128 + (16 * fq + 8 * ks) * 2);
        }
    }

    // input patch of a tile (19 rows x 105 floats), fetched into registers one tile ahead (the HBM latency hides behind the
    // previous tile's two MFMA phases): four patch rows per pass, 128 threads per row (105 used), so a thread's column, pixel
    // and channel never change and its row advances by 4 -- no division, LDS offsets are immediates
    constexpr int NIN = (INY + 3) / 4;
    float in_r[NIN];
    const int in_col = tid & 127, in_row0 = tid >> 7;
    const bool in_col_ok = in_col < INX * 3;
    const int in_px = in_col / 3;
    T *const in_dst = sIn + in_row0 * IN_LD + in_pix(in_px) + (in_col - 3 * in_px);
    auto tile_origin = [&](int tile, int &n, int &oy0, int &ox0) {
        const uint32_t tyx = fdiv((uint32_t)tile, p.dtx);
        const int tx = (int)((uint32_t)tile - tyx * (uint32_t)p.tiles_x);
        n = (int)fdiv(tyx, p.dty);
        const int ty = (int)(tyx - (uint32_t)n * (uint32_t)p.tiles_y);
        oy0 = ty * TY; ox0 = tx * TX;
    };
    auto fetch_input = [&](int tile) {
        int n, oy0, ox0;
        tile_origin(tile, n, oy0, ox0);
        const float *img = p.in + (long long)n * p.in_img_stride;
        const int gy0 = 2 * oy0 - 2 + in_row0, gx3 = (2 * ox0 - 2) * 3 + in_col;
        const bool xok = in_col_ok && (unsigned)gx3 < (unsigned)(3 * p.W);
        const int w3 = 3 * p.W;
#pragma unroll
        for (int it = 0; it < NIN; ++it) {
            const int gy = gy0 + 4 * it;
            const bool ok = xok && (unsigned)gy < (unsigned)p.H && (4 * it + in_row0 < INY);
            const float v = img[ok ? gy * w3 + gx3 : 0];      // always-valid address, then select (an image is < 2^31 floats)
            in_r[it] = ok ? v : 0.f;
        }
    };

    int tile = blockIdx.x;
    fetch_input(tile);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // weights landed (once), first input patch in registers
    __syncthreads();                                    // the zero fill of the input patch is ordered before the first tile's stores
    for (; tile < p.n_tiles; tile += gridDim.x) {
        int n, oy0, ox0;
        tile_origin(tile, n, oy0, ox0);
        const int y1_0 = 2 * oy0 - 1, x1_0 = 2 * ox0 - 1;      // layer-1 position of patch element (0, 0)
        // float32 -> fp16 (same operand rounding as the unfused first-layer kernel)
#pragma unroll
        for (int it = 0; it < NIN; ++it)
            if (in_col_ok && 4 * it + in_row0 < INY) in_dst[4 * it * IN_LD] = (T)in_r[it];
        __syncthreads();    // input patch visible; every wave is past phase 2 of the previous tile (patch P is free)
        if (tile + (int)gridDim.x < p.n_tiles) fetch_input(tile + gridDim.x);

        // ---- phase 1: layer 1 for the 17 x 33 positions --------------------------------------------
        // A group of 16 positions has ONE column parity (groups 0..18: the 289 even-column positions, 19..35: one row of 16 odd columns
        // each), so its 16 lanes store 16 CONSECUTIVE rows of the de-interleaved patch: rowP = 16 g + fr.
        for (int g = wave; g < NGRP; g += 8) {
            const int rowP = g * 16 + fr;
            const bool live = g >= GEVEN || rowP < NEVEN;
            int py, px;
            if (g < GEVEN) {            // (wave-uniform)
                const int e = live ? rowP : NEVEN - 1;
                py = e / EVEN_COLS;
                px = 2 * (e - py * EVEN_COLS);
            } else {
                py = g - GEVEN;
                px = 2 * fr + 1;
            }
            const T *src = sIn + py * IN_LD;
            const int q0 = in_pix(px + 2 * (fq & 1)), q1 = in_pix(px + 2 * (fq & 1) + 1);
            typedef unsigned long long u64;
            const u64 b00 = *reinterpret_cast<const u64 *>(src + brow0 + q0), b01 = *reinterpret_cast<const u64 *>(src + brow0 + q1);
            const u64 b10 = *reinterpret_cast<const u64 *>(src + brow1 + q0), b11 = *reinterpret_cast<const u64 *>(src + brow1 + q1);
            uint4v b0, b1;
            b0.x = (unsigned)b00; b0.y = (unsigned)(b00 >> 32); b0.z = (unsigned)b01; b0.w = (unsigned)(b01 >> 32);
            b1.x = (unsigned)b10; b1.y = (unsigned)(b10 >> 32); b1.z = (unsigned)b11; b1.w = (unsigned)(b11 >> 32);
            float4v d0 = mma_chunk<T>(a1[0][0], b0, bias1[0]);
            float4v d1 = mma_chunk<T>(a1[1][0], b0, bias1[1]);
            d0 = mma_chunk<T>(a1[0][1], b1, d0);
            d1 = mma_chunk<T>(a1[1][1], b1, d1);
            // lane: channels 8 fq + {0..3} (d0) and 8 fq + {4..7} (d1) of position fr
            const int gy = y1_0 + py, gx = x1_0 + px;
            const bool inside = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            T o[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v0 = leaky01(d0[j]), v1 = leaky01(d1[j]);
                o[j] = (T)v0;
                o[4 + j] = (T)v1;
            }
            uint4v u;
            __builtin_memcpy(&u, o, 16);
            if (!inside) u = uint4v{0u, 0u, 0u, 0u};        // layer 2's zero padding (tiles on the image border only)
            if (live) *reinterpret_cast<uint4v *>(sP + rowP * 64 + ((fq ^ stem_swz_p(rowP)) << 4)) = u;
        }
        __syncthreads();    // patch P complete; the input patch may be overwritten

        // ---- phase 2: layer 2, wave = output row oy0 + wave, 16 pixels x 64 couts ------------------
        float4v acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = *reinterpret_cast<const float4v *>(sBias + fq * 16 + 4 * a);     // bias as MFMA C input
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {        // three pixel fragments at a time: their latency overlaps
            uint4v fb[3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int rowP = (kw & 1) ? ODD0 + (2 * wave + kh) * ODD_COLS + fr : (2 * wave + kh) * EVEN_COLS + fr + (kw >> 1);
                fb[kw] = *reinterpret_cast<const uint4v *>(sP + rowP * 64 + ((fq ^ stem_swz_p(rowP)) << 4));
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                uint4v fa[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) fa[a] = *reinterpret_cast<const uint4v *>(sW + (kh * 3 + kw) * 4096 + a * 1024 + a_frag);
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = mma_chunk<T>(fa[a], fb[kw], acc[a]);
            }
        }
        const int oy = oy0 + wave, ox = ox0 + fr;
        const bool valid = oy < p.Ho && ox < p.Wo;
        T o[16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = acc[a][j];
                o[4 * a + j] = (T)leaky01(v);
            }
        uint4v u0, u1;
        __builtin_memcpy(&u0, o, 16);
        __builtin_memcpy(&u1, o + 8, 16);
        const long long pix = (long long)oy * p.Wo + ox;
        if (valid) {
            T *op = reinterpret_cast<T *>(p.out) + (long long)n * p.out_img_stride + pix * p.out_ld + fq * 16;
            *reinterpret_cast<uint4v *>(op) = u0;
            *reinterpret_cast<uint4v *>(op + 8) = u1;
        }
        if (p.w3) {         // (wave-uniform) layer 3 on the same 16 pixels; operands = the fp16 values just stored
            float4v d0 = mma_chunk<T>(a3[0][0], u0, *reinterpret_cast<const float4v *>(sBias + 64 + 8 * fq));
            float4v d1 = mma_chunk<T>(a3[1][0], u0, *reinterpret_cast<const float4v *>(sBias + 64 + 8 * fq + 4));
            d0 = mma_chunk<T>(a3[0][1], u1, d0);
            d1 = mma_chunk<T>(a3[1][1], u1, d1);
            T o3[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v0 = d0[j], v1 = d1[j];
                o3[j] = (T)leaky01(v0);
                o3[4 + j] = (T)leaky01(v1);
            }
            uint4v u3;
            __builtin_memcpy(&u3, o3, 16);
            if (valid)
                *reinterpret_cast<uint4v *>(reinterpret_cast<T *>(p.out3) + (long long)n * p.out3_img_stride + pix * p.out3_ld + fq * 8) = u3;
        }
    }
}

hipError_t launch_stem(const StemParams &p0, int batch, hipStream_t s, int max_grid) {
    StemParams p = p0;
    if ((p.H & 1) || (p.W & 1) || p.Ho != p.H / 2 || p.Wo != p.W / 2) return hipErrorInvalidValue;
    p.tiles_x = (p.Wo + TX - 1) / TX;
    p.tiles_y = (p.Ho + TY - 1) / TY;
    p.dtx = make_fastdiv((uint32_t)p.tiles_x);
    p.dty = make_fastdiv((uint32_t)p.tiles_y);
    const long long tiles = (long long)batch * p.tiles_x * p.tiles_y;
    if (tiles <= 0 || tiles > 0x7fffffffLL) return hipErrorInvalidValue;
    p.n_tiles = (int)tiles;
    // persistent workgroups: two per CU (78 KB of LDS each), every one walks tiles blockIdx, blockIdx + grid, ...
    const unsigned grid = (unsigned)(tiles < max_grid ? tiles : max_grid);
    hipLaunchKernelGGL(stem_v3_kernel, dim3(grid), dim3(512), 0, s, p);
    return hipGetLastError();
}

}  // namespace yolo
