// First layer: 3x3 stride-1 conv on the 3-channel input (+ folded BN + leaky).  K = 27 is too thin
// for the implicit-GEMM tiling (the generic kernel spent 1 ms here at 20 TFLOP/s, 11 % of a YOLOv3
// step) and the layer is bound by its 64-byte-per-pixel output stream, so this is a direct
// convolution on the vector ALU: one thread = one output pixel x all couts, weights as scalar (SGPR)
// operands of packed FMAs, output transposed through LDS so each store instruction writes 1 KiB of
// contiguous NHWC, float32 input read straight from the
// caller's tensor (this replaces the input cast/pad pass as well: reference net/layers.py:106-109
// placeholder + :17-67 conv2d_bn_act), coalesced 16-byte NHWC stores.
// fp16 nets: input and weights are rounded to fp16 first (same operands as the MFMA path), products
// accumulate in fp32.
#include "conv_common.h"
#include <type_traits>

namespace yolo {

// POOL: the 2x2 / stride-2 max-pool that follows the first conv in Darknet-19 and tiny-YOLO (net/v2.py) is taken in
// registers: a wave covers 32 x-positions of two adjacent rows (lane = 32 * row + x), the 2x2 window is a max over
// lanes l ^ 1 and l ^ 32 of the raw accumulators (bias and leaky are monotone: pooling first is the same result with
// a quarter of the activation work), and only the pooled pixel row (16 per wave) is written.  H and W even.
template <bool F32, int COUT, bool POOL>
__global__ void __launch_bounds__(256) conv_first_kernel(const FirstParams p) {
    typedef typename std::conditional<F32, float, _Float16>::type T;
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int ROWB = COUT * (int)sizeof(T);     // bytes of one output pixel
    constexpr int NCH = ROWB / 16;                  // 16-byte chunks per pixel
    constexpr int LSTR = ROWB + 16;                 // padded LDS row: conflict-free ds_write_b128 for 8 consecutive lanes
    __shared__ __attribute__((aligned(16))) unsigned char stage[4 * 64 * LSTR];
    // weights/bias are read at wave-uniform addresses through the constant address space: scalar
    // loads into SGPRs (scalar-cache resident, 3.5 KB) used as the SGPR operand of the packed FMAs
    typedef const __attribute__((address_space(4))) float cfloat;
    cfloat *wg = (cfloat *)p.wgt;
    cfloat *bg = (cfloat *)p.bias;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    unsigned char *mine = stage + wave * (64 * LSTR);

    long long px0 = 0, n;
    int ox, oy;
    bool live;
    int pool_x0 = 0, pool_yp = 0;
    if (POOL) {
        // one workgroup = 2 rows x 128 x-positions: blockIdx -> (image, pooled row, x block)
        const uint32_t t = fdiv(blockIdx.x, p.dXB);
        const int xb = (int)(blockIdx.x - t * (uint32_t)p.xblocks);
        n = fdiv(t, p.dHp);
        pool_yp = (int)(t - (uint32_t)n * (uint32_t)(p.H >> 1));
        pool_x0 = xb * 128 + wave * 32;
        ox = pool_x0 + (lane & 31);
        oy = 2 * pool_yp + (lane >> 5);
        live = ox < p.W;
        if (!live) ox = p.W - 1;
    } else {
        // one workgroup = 256 consecutive output pixels (n, oy, ox) in NHWC order; p.total = B*H*W
        px0 = (long long)blockIdx.x * 256 + wave * 64;      // first pixel of this wave
        const long long pix = px0 + lane;
        live = pix < p.total;
        const long long pp = live ? pix : p.total - 1;
        const uint32_t t = fdiv((uint32_t)pp, p.dW);         // p.total < 2^31 (checked by the launcher)
        ox = (int)((uint32_t)pp - t * (uint32_t)p.W);
        n = fdiv(t, p.dH);
        oy = (int)(t - (uint32_t)n * (uint32_t)p.H);
    }

    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = bg[co];
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy - 1 + kh;                         // SAME padding: one zero row/col each side
        const bool rowok = (unsigned)iy < (unsigned)p.H;
        const float *rowp = p.in + ((n * p.H + (rowok ? iy : 0)) * (long long)p.W) * 3;
        float xin[3][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int ix = ox - 1 + c;
            const bool ok = rowok && (unsigned)ix < (unsigned)p.W;
            const int ixc = ix < 0 ? 0 : (ix >= p.W ? p.W - 1 : ix);    // always-valid address: load, then select
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                float v = rowp[ixc * 3 + ci];
                if (p.round_half) v = (float)(_Float16)v;
                xin[c][ci] = ok ? v : 0.f;
            }
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                cfloat *w = wg + ((kh * 3 + kw) * 3 + ci) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xin[kw][ci], w[co], acc[co]);
            }
    }
    if (POOL) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float v = live ? acc[co] : -3.0e38f;
            v = fmaxf(v, __shfl_xor(v, 1));
            v = fmaxf(v, __shfl_xor(v, 32));
            acc[co] = v;
        }
    }
    // leaky, convert, transpose through LDS: lane -> its pixel's row; then every store instruction of
    // the wave writes 1 KiB of contiguous NHWC output (16 pixels x ROWB)
    T tv[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float v = acc[co];
        if (p.leaky) v = fmaxf(0.1f * v, v);
        tv[co] = (T)v;
    }
    const int slot = POOL ? (lane & 31) >> 1 : lane;       // LDS row of this lane's (pooled) pixel
    if (!POOL || ((lane & 33) == 0)) {                       // POOL: even x of the upper row holds the window maximum
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            uint4v u;
            __builtin_memcpy(&u, tv + q * EPC, 16);
            *reinterpret_cast<uint4v *>(mine + slot * LSTR + q * 16) = u;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): the wave's own LDS writes are done (wave-private tile)
    __builtin_amdgcn_wave_barrier();
    T *obase = reinterpret_cast<T *>(p.out);
    if (POOL) {
        const int Wp = p.W >> 1, Hp = p.H >> 1;
#pragma unroll
        for (int k = 0; k < (16 * NCH + 63) / 64; ++k) {
            const int idx = k * 64 + lane;      // chunk index inside the wave's 16 pooled pixels
            const int lp = idx / NCH, ch = idx % NCH;
            const int pxg = (pool_x0 >> 1) + lp;
            if (idx < 16 * NCH && pxg < Wp) {
                const uint4v u = *reinterpret_cast<const uint4v *>(mine + lp * LSTR + ch * 16);
                const long long rem = (long long)pool_yp * Wp + pxg;
                *reinterpret_cast<uint4v *>(obase + n * p.out_img_stride + rem * p.out_ld + ch * EPC) = u;
            }
        }
        (void)Hp;
        return;
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int idx = k * 64 + lane;          // chunk index inside the wave's 64-pixel tile
        const int lp = idx / NCH, ch = idx % NCH;
        const long long gp = px0 + lp;
        if (gp < p.total) {
            const uint4v u = *reinterpret_cast<const uint4v *>(mine + lp * LSTR + ch * 16);
            const long long img = fdiv((uint32_t)gp, p.dHW);
            const long long rem = gp - img * (long long)p.H * p.W;
            *reinterpret_cast<uint4v *>(obase + img * p.out_img_stride + rem * p.out_ld + ch * EPC) = u;
        }
    }
}

// ---- fp16 nets, 32 couts, fused pool: the same layer on the MATRIX cores ------------------------------------------------------
// The direct VALU kernel above runs Darknet-19's first layer + pool (416 x 416, batch 16) in 94 us = 10 % of the YOLOv2 step at
// 0.8 TB/s: 864 FMAs per pixel make it VALU-bound.  This is the layer-1 phase of stem.hip on its own: a persistent workgroup (8 waves)
// walks tiles of 8 x 16 POOLED outputs = 16 x 32 conv positions; the 18 x 34 x 3 float32 input patch goes, one tile ahead through
// registers, into LDS as 8-byte fp16 pixels (R G B 0); a lane's B fragment = two whole neighbouring pixels of one patch row (two
// ds_read_b64), K = 27 spread over two 32-deep k-steps, bias as the MFMA's C input; wave w owns pooled row w: conv rows 2w, 2w + 1 are
// two fragments of the same lane (vertical max in registers), the columns of a window are lanes fr, fr ^ 1; leaky after the pool
// (monotone: same value), one 16-byte store per lane (8 couts of a pooled pixel; the four lane groups make its 64 bytes).
namespace {
constexpr int FM_TY = 8, FM_TX = 16;                 // pooled outputs per tile
constexpr int FM_INY = 2 * FM_TY + 2, FM_INX = 2 * FM_TX + 2;       // 18 x 34 input pixels
constexpr int FM_PX = 36;                            // pixels per LDS patch row (34 used, index 34 only meets zero weights)
constexpr int FM_LD = FM_PX * 4;                     // halfs per patch row
}  // namespace

__global__ void __launch_bounds__(512, 4) first_pool_mfma_kernel(const FirstParams p) {
    typedef _Float16 T;
    __shared__ __attribute__((aligned(16))) unsigned char smem[FM_INY * FM_LD * 2];
    T *const sIn = reinterpret_cast<T *>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // layer weights as MFMA A fragments (see stem.hip): tile t, row fr holds cout 8*(fr>>2) + 4t + (fr&3); k-step s, lane group fq,
    // element j -> patch row kh = (s == 0 ? fq >> 1 : 2), pixel pw = 2 (fq & 1) + (j >> 2), channel c = j & 3
    uint4v a1[2][2];
    float4v bias1[2];
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
                const float w = p.wgt[(live ? (kh * 3 + pw) * 3 + c : 0) * 32 + ch];
                h[j] = live ? (T)w : (T)0.f;
            }
            __builtin_memcpy(&a1[t][st], h, 16);
        }
    }
    bias1[0] = *reinterpret_cast<const float4v *>(p.bias + 8 * fq);
    bias1[1] = *reinterpret_cast<const float4v *>(p.bias + 8 * fq + 4);
    for (int i = tid; i < FM_INY * FM_PX; i += 512) *reinterpret_cast<unsigned long long *>(sIn + i * 4) = 0ull;   // 4th half / spare pixels: zero for good
    const int boff0 = (fq >> 1) * FM_LD + 2 * (fq & 1) * 4;
    const int boff1 = (fq < 2 ? 2 : (fq >> 1)) * FM_LD + 2 * (fq & 1) * 4;

    // input staging: four patch rows per pass, 128 threads per row (102 used)
    constexpr int NIN = (FM_INY + 3) / 4;
    float in_r[NIN];
    const int in_col = tid & 127, in_row0 = tid >> 7;
    const bool in_col_ok = in_col < FM_INX * 3;
    const int in_px = in_col / 3;
    T *const in_dst = sIn + in_row0 * FM_LD + in_px * 4 + (in_col - 3 * in_px);
    const int Hp = p.H >> 1, Wp = p.W >> 1;
    auto tile_origin = [&](int tile, int &n, int &py0, int &px0) {
        const uint32_t tyx = fdiv((uint32_t)tile, p.dXB);
        const int tx = (int)((uint32_t)tile - tyx * (uint32_t)p.xblocks);
        n = (int)fdiv(tyx, p.dHp);
        const int ty = (int)(tyx - (uint32_t)n * (uint32_t)p.tiles_y);
        py0 = ty * FM_TY; px0 = tx * FM_TX;
    };
    auto fetch_input = [&](int tile) {
        int n, py0, px0;
        tile_origin(tile, n, py0, px0);
        const float *img = p.in + (long long)n * p.H * p.W * 3;
        const int gy0 = 2 * py0 - 1 + in_row0, gx3 = (2 * px0 - 1) * 3 + in_col;
        const bool xok = in_col_ok && (unsigned)gx3 < (unsigned)(3 * p.W);
        const int w3 = 3 * p.W;
#pragma unroll
        for (int it = 0; it < NIN; ++it) {
            const int gy = gy0 + 4 * it;
            const bool ok = xok && (unsigned)gy < (unsigned)p.H && (4 * it + in_row0 < FM_INY);
            const float v = img[ok ? gy * w3 + gx3 : 0];
            in_r[it] = ok ? v : 0.f;
        }
    };

    int tile = blockIdx.x;
    fetch_input(tile);
    __syncthreads();            // the zero fill is ordered before the first tile's stores
    for (; tile < p.n_tiles; tile += gridDim.x) {
        int n, py0, px0;
        tile_origin(tile, n, py0, px0);
#pragma unroll
        for (int it = 0; it < NIN; ++it)
            if (in_col_ok && 4 * it + in_row0 < FM_INY) in_dst[4 * it * FM_LD] = (T)in_r[it];
        __syncthreads();        // input patch visible
        if (tile + (int)gridDim.x < p.n_tiles) fetch_input(tile + gridDim.x);
        const int oy = py0 + wave;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4v m0, m1;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const T *src = sIn + (2 * wave + r) * FM_LD + (16 * half + fr) * 4;
                typedef unsigned long long u64;
                const u64 b00 = *reinterpret_cast<const u64 *>(src + boff0), b01 = *reinterpret_cast<const u64 *>(src + boff0 + 4);
                const u64 b10 = *reinterpret_cast<const u64 *>(src + boff1), b11 = *reinterpret_cast<const u64 *>(src + boff1 + 4);
                uint4v b0, b1;
                b0.x = (unsigned)b00; b0.y = (unsigned)(b00 >> 32); b0.z = (unsigned)b01; b0.w = (unsigned)(b01 >> 32);
                b1.x = (unsigned)b10; b1.y = (unsigned)(b10 >> 32); b1.z = (unsigned)b11; b1.w = (unsigned)(b11 >> 32);
                float4v d0 = mma_chunk<T>(a1[0][0], b0, bias1[0]);
                float4v d1 = mma_chunk<T>(a1[1][0], b0, bias1[1]);
                d0 = mma_chunk<T>(a1[0][1], b1, d0);
                d1 = mma_chunk<T>(a1[1][1], b1, d1);
                if (r == 0) { m0 = d0; m1 = d1; }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { m0[j] = fmaxf(m0[j], d0[j]); m1[j] = fmaxf(m1[j], d1[j]); }
                }
            }
            T o[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v0 = fmaxf(m0[j], __shfl_xor(m0[j], 1)), v1 = fmaxf(m1[j], __shfl_xor(m1[j], 1));     // columns ox, ox ^ 1
                if (p.leaky) { v0 = fmaxf(0.1f * v0, v0); v1 = fmaxf(0.1f * v1, v1); }
                o[j] = (T)v0;
                o[4 + j] = (T)v1;
            }
            const int ox = px0 + 8 * half + (fr >> 1);
            if (!(fr & 1) && oy < Hp && ox < Wp) {
                uint4v u;
                __builtin_memcpy(&u, o, 16);
                T *op = reinterpret_cast<T *>(p.out) + (long long)n * p.out_img_stride + ((long long)oy * Wp + ox) * p.out_ld + 8 * fq;
                *reinterpret_cast<uint4v *>(op) = u;
            }
        }
        __syncthreads();        // every wave has read its patch rows: the next tile may overwrite them
    }
}

// ---- float32 nets, fused pool: the same on the float32 matrix instruction --------------------------------------------------------
// tiny-YOLOv2-VOC's first layer (3 -> 16 + pool, 416 x 416, batch 64) ran 218 us on the VALU kernel: 1.4 TB/s where its 310 MB allow
// ~70 us.  mfma_f32_16x16x4f32 takes ONE float per lane and operand: k = 4 s + (lane >> 4) for k-step s, so K = 27 is seven k-steps
// with a single wasted slot, a lane's B value is one ds_read_b32 at a loop-invariant offset (kh, kw, c) from its pixel, the A values
// are seven registers per 16 couts, and the bias is the C input.  Same tiling as above: persistent workgroups, 8 x 16 pooled outputs
// per tile, wave w = pooled row w (conv rows 2 w, 2 w + 1 = two accumulators of a lane, columns = lanes fr, fr ^ 1), the input patch
// as 16-byte (R G B 0) float32 pixels in LDS, fetched one tile ahead through registers.  NT = Cout / 16.
template <int NT>
__global__ void __launch_bounds__(512, 4) first_pool_mfma_f32_kernel(const FirstParams p) {
    constexpr int LD = FM_PX * 4;               // floats per patch row
    __shared__ __attribute__((aligned(16))) float sIn[FM_INY * LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int COUT = 16 * NT;

    float a1[NT][7];
    int koff[7];
    float4v bias1[NT];
#pragma unroll
    for (int st = 0; st < 7; ++st) {
        const int k = 4 * st + fq;              // (kh, kw, c), c fastest: the K_FIRST packing of the weights
        const bool live = k < 27;
        const int kk = live ? k : 0;
        const int kh = kk / 9, kw = (kk - 9 * kh) / 3, c = kk - 9 * kh - 3 * kw;
        koff[st] = (kh * FM_PX + kw) * 4 + c;   // a dead slot re-reads element 0: finite, its weight is zero
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float w = p.wgt[kk * COUT + 16 * t + fr];
            a1[t][st] = live ? w : 0.f;
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) bias1[t] = *reinterpret_cast<const float4v *>(p.bias + 16 * t + 4 * fq);       // rows 4 fq + j of fragment t
    for (int i = tid; i < FM_INY * FM_PX; i += 512) *reinterpret_cast<float4v *>(sIn + i * 4) = float4v{0.f, 0.f, 0.f, 0.f};

    constexpr int NIN = (FM_INY + 3) / 4;
    float in_r[NIN];
    const int in_col = tid & 127, in_row0 = tid >> 7;
    const bool in_col_ok = in_col < FM_INX * 3;
    const int in_px = in_col / 3;
    float *const in_dst = sIn + in_row0 * LD + in_px * 4 + (in_col - 3 * in_px);
    const int Hp = p.H >> 1, Wp = p.W >> 1;
    auto tile_origin = [&](int tile, int &n, int &py0, int &px0) {
        const uint32_t tyx = fdiv((uint32_t)tile, p.dXB);
        const int tx = (int)((uint32_t)tile - tyx * (uint32_t)p.xblocks);
        n = (int)fdiv(tyx, p.dHp);
        const int ty = (int)(tyx - (uint32_t)n * (uint32_t)p.tiles_y);
        py0 = ty * FM_TY; px0 = tx * FM_TX;
    };
    auto fetch_input = [&](int tile) {
        int n, py0, px0;
        tile_origin(tile, n, py0, px0);
        const float *img = p.in + (long long)n * p.H * p.W * 3;
        const int gy0 = 2 * py0 - 1 + in_row0, gx3 = (2 * px0 - 1) * 3 + in_col;
        const bool xok = in_col_ok && (unsigned)gx3 < (unsigned)(3 * p.W);
        const int w3 = 3 * p.W;
#pragma unroll
        for (int it = 0; it < NIN; ++it) {
            const int gy = gy0 + 4 * it;
            const bool ok = xok && (unsigned)gy < (unsigned)p.H && (4 * it + in_row0 < FM_INY);
            const float v = img[ok ? gy * w3 + gx3 : 0];
            in_r[it] = ok ? v : 0.f;
        }
    };

    int tile = blockIdx.x;
    fetch_input(tile);
    __syncthreads();            // the zero fill is ordered before the first tile's stores
    for (; tile < p.n_tiles; tile += gridDim.x) {
        int n, py0, px0;
        tile_origin(tile, n, py0, px0);
#pragma unroll
        for (int it = 0; it < NIN; ++it)
            if (in_col_ok && 4 * it + in_row0 < FM_INY) in_dst[4 * it * LD] = in_r[it];
        __syncthreads();        // input patch visible
        if (tile + (int)gridDim.x < p.n_tiles) fetch_input(tile + gridDim.x);
        const int oy = py0 + wave;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4v m[NT];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float *src = sIn + (2 * wave + r) * LD + (16 * half + fr) * 4;
                float b[7];
#pragma unroll
                for (int st = 0; st < 7; ++st) b[st] = src[koff[st]];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float4v d = bias1[t];
#pragma unroll
                    for (int st = 0; st < 7; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t][st], b[st], d, 0, 0, 0);
                    if (r == 0) m[t] = d;
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) m[t][j] = fmaxf(m[t][j], d[j]);
                    }
                }
            }
            const int ox = px0 + 8 * half + (fr >> 1);
            const bool store = !(fr & 1) && oy < Hp && ox < Wp;
            float *op = reinterpret_cast<float *>(p.out) + (long long)n * p.out_img_stride + ((long long)oy * Wp + ox) * p.out_ld + 4 * fq;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float4v v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float x = fmaxf(m[t][j], __shfl_xor(m[t][j], 1));       // columns ox, ox ^ 1
                    if (p.leaky) x = fmaxf(0.1f * x, x);
                    v[j] = x;
                }
                if (store) *reinterpret_cast<float4v *>(op + 16 * t) = v;
            }
        }
        __syncthreads();        // every wave has read its patch rows: the next tile may overwrite them
    }
}

// the matrix-core forms of layer 1 + pool: fp16 nets with 32 couts, float32 nets with 16 or 32 (16-byte aligned float32 output view)
static bool first_mfma_applies(int dtype, int cout, bool pool) {
    return pool && (dtype == YOLO_DTYPE_F16 ? cout == 32 : (cout == 16 || cout == 32));
}

template <bool T, int COUT>
static void launch_first_t(const FirstParams &p, dim3 grid, hipStream_t s) {
    if (p.pool) hipLaunchKernelGGL((conv_first_kernel<T, COUT, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_first_kernel<T, COUT, false>), grid, dim3(256), 0, s, p);
}

hipError_t launch_first(const FirstParams &p0, int dtype, hipStream_t s) {
    FirstParams p = p0;
    long long g = (p.total + 255) / 256;            // one workgroup per 256 output pixels
    if (g < 1 || p.total > 0x7fffffffLL) return hipErrorInvalidValue;   // 32-bit pixel indices (fdiv)
    p.dW = make_fastdiv((uint32_t)p.W);
    p.dH = make_fastdiv((uint32_t)p.H);
    p.dHW = make_fastdiv((uint32_t)p.H * (uint32_t)p.W);
    if (p.pool) {                                   // one workgroup per (image, pooled row, 128-wide x block)
        if ((p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;
        p.xblocks = (p.W + 127) / 128;
        p.dXB = make_fastdiv((uint32_t)p.xblocks);
        p.dHp = make_fastdiv((uint32_t)(p.H >> 1));
        g = p.total / ((long long)p.H * p.W) * (p.H >> 1) * p.xblocks;
        if (g < 1 || g > 0x7fffffffLL) return hipErrorInvalidValue;
    }
    if (first_mfma_applies(dtype, p.Cout, p.pool != 0)) {        // fp16, 32 couts, fused pool: the MFMA form
        const int Hp = p.H >> 1, Wp = p.W >> 1;
        p.xblocks = (Wp + FM_TX - 1) / FM_TX;
        p.tiles_y = (Hp + FM_TY - 1) / FM_TY;
        p.dXB = make_fastdiv((uint32_t)p.xblocks);
        p.dHp = make_fastdiv((uint32_t)p.tiles_y);
        const long long tiles = p.total / ((long long)p.H * p.W) * p.xblocks * p.tiles_y;
        if (tiles < 1 || tiles > 0x7fffffffLL || (long long)p.H * p.W * 3 > 0x7fffffffLL) return hipErrorInvalidValue;
        p.n_tiles = (int)tiles;
        const dim3 mgrid((unsigned)(tiles < 1024 ? tiles : 1024));
        if (dtype == YOLO_DTYPE_F16) hipLaunchKernelGGL(first_pool_mfma_kernel, mgrid, dim3(512), 0, s, p);
        else if (p.Cout == 16) hipLaunchKernelGGL(first_pool_mfma_f32_kernel<1>, mgrid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL(first_pool_mfma_f32_kernel<2>, mgrid, dim3(512), 0, s, p);
        return hipGetLastError();
    }
    const dim3 grid((unsigned)g);
    if (dtype == YOLO_DTYPE_F16) {
        if (p.Cout == 32) launch_first_t<false, 32>(p, grid, s);
        else if (p.Cout == 16) launch_first_t<false, 16>(p, grid, s);
        else return hipErrorInvalidValue;
    } else {
        if (p.Cout == 32) launch_first_t<true, 32>(p, grid, s);
        else if (p.Cout == 16) launch_first_t<true, 16>(p, grid, s);
        else return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

std::string first_symbol(int dtype, int cout, bool pool) {
    if (first_mfma_applies(dtype, cout, pool))
        return dtype == YOLO_DTYPE_F16 ? "yolo::first_pool_mfma_kernel(yolo::FirstParams)"
                                       : std::string("void yolo::first_pool_mfma_f32_kernel<") + std::to_string(cout / 16) + ">(yolo::FirstParams)";
    return std::string("void yolo::conv_first_kernel<") + (dtype == YOLO_DTYPE_F16 ? "false" : "true") + ", " + std::to_string(cout) + ", " +
           (pool ? "true" : "false") + ">(yolo::FirstParams)";
}

}  // namespace yolo
