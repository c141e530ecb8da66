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
#include "yolo_internal.h"
#include <type_traits>

namespace yolo {

typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

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
    return std::string("void yolo::conv_first_kernel<") + (dtype == YOLO_DTYPE_F16 ? "false" : "true") + ", " + std::to_string(cout) + ", " +
           (pool ? "true" : "false") + ">(yolo::FirstParams)";
}

}  // namespace yolo
