// Internal interfaces of libyolo_hip.so (not part of the C ABI; see include/yolo_hip.h).
//
// Data layout in HBM (DESIGN.md "Layout"):
//   activations  NHWC, element type T (fp16 or fp32), addressed as strided views
//                (pixel stride `ld`, first channel `coff`, image stride) so that
//                route/concat, upsample and reorg never copy: producers write slices.
//   weights      [Cout_pad][K] with K = (kh, kw, cin) in 16-byte chunks, BN folded,
//                rows zero-padded to a multiple of 128, K zero-padded to 8 chunks.
//   head logits  float32 in the reference's layout (net/v2.py:52-59, net/layers.py:119-133).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "yolo_hip.h"

namespace yolo {

enum OutMode { OUT_NORMAL = 0, OUT_UP2 = 1, OUT_REORG2 = 2, OUT_POOL2 = 3 };   // OUT_POOL2: the 2x2/2 max-pool behind the conv is taken in its epilogue (2-D tap tiles)
enum BufId { BUF_NONE = -1, BUF_USER_OUT = -2, BUF_USER_IN = -3 };
enum ConvCfg { CFG_N128 = 0, CFG_N64 = 1, CFG_N32 = 2 };   // cout-tile width of the block
enum KernelKind { K_PREP = 0, K_CONV = 1, K_POOL = 2, K_ELTWISE = 3, K_FIRST = 4 };

// A strided NHWC view inside a planned buffer.
struct View {
    int buf = BUF_NONE;
    int H = 0, W = 0, C = 0;
    int ld = 0;                 // elements between consecutive pixels
    int coff = 0;               // first channel inside the pixel
    long long img_stride = 0;   // elements between consecutive images
    long long base = 0;         // extra element offset (head scale offset inside the output)
    bool f32 = false;           // float32 elements whatever the net dtype (head logits)
};

struct Buffer {
    long long elems_per_image = 0;  // elements per image
    int esize = 2;
    int first = 1 << 30, last = -1; // kernel indices of first write / last access
    size_t offset = 0;              // bytes inside the workspace
    size_t bytes = 0;               // for max_batch
    size_t used = 0;                // payload bytes of the region (the rest is alignment slack + yolo_net_options.guard_bytes)
    bool is_concat = false;
};

// ---- device-side parameter blocks -----------------------------------------------------
// Division by a launch-time constant without the ~40-instruction software divide (Granlund-Montgomery,
// exact for every 32-bit unsigned n): q = (t + ((n - t) >> sh1)) >> sh2 with t = mulhi(mul, n).
struct FastDiv {
    uint32_t mul, sh1, sh2, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{1, 0, 0, d ? d : 1};
    uint32_t l = 0;
    while ((1ull << l) < f.d) ++l;
    f.mul = (uint32_t)((((1ull << l) - f.d) << 32) / f.d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 0 ? l - 1 : 0;
    return f;
}
__host__ __device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(f.mul, n);
#else
    const uint32_t t = (uint32_t)(((unsigned long long)f.mul * n) >> 32);
#endif
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

// leaky ReLU 0.1 (net/layers.py:6,50-51: tf.nn.leaky_relu = max(alpha x, x)) as TWO instructions: fmaxf() on an MFMA result
// makes hipcc put a canonicalising `v_max x, x` in front of the real one (three instructions per value).  Used where the kernel is VALU-bound (stem.hip:
// -5 %); the conv epilogues keep fmaxf(), whose instructions the compiler schedules freely (the opaque asm cost them +1 %,
// the staged float32 head epilogue +19 %: profiles/r03_ablation.md).  Same value for every non-NaN input.
__host__ __device__ __forceinline__ float leaky01(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    const float y = 0.1f * x;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(y), "v"(x));
    return r;
#else
    return x > 0.1f * x ? x : 0.1f * x;
#endif
}

// The per-image candidate counters of the decode live one per 128-byte line: with the 32 counters of a batch in ONE line every
// candidate's atomicAdd of the whole batch went through one L2 channel (YOLOv3-608 b32: 2 900 atomics, decode_kernel 31 us, 5 us with
// no candidate at all).
constexpr int kCandCountStride = 32;

struct ConvParams {
    const void *in;            // base of the input BUFFER (view offsets are folded into byte offsets)
    const void *wgt;
    const float *bias;
    const void *res;           // residual tensor base (element pointer incl. coff) or null
    void *out;                 // output base pointer incl. view base / coff
    uint32_t in_bytes, wgt_bytes;
    uint32_t out_bytes, res_bytes;     // extent of the output / residual tensors from `out` / `res` (0: beyond 2 GiB -> no buffer addressing)
    int H, W, in_ld, in_coff;
    long long in_img_stride;
    int Ho, Wo, HoWo, M;
    int Cout, out_ld;
    long long out_img_stride;
    int res_ld;
    long long res_img_stride;
    int ksize, stride, pad, taps;
    int ktiles, tiles_per_tap, cpt_shift;
    int cin_chunks;            // 16-byte chunks per tap (storage channels / EPC)
    uint32_t wrow_bytes;
    int leaky, has_res, outmode, out_f32, vec_out, vec_res;
    int f32;                   // elements of input / weights / residual are float32 (else fp16)
    int n_tiles_n, n_blocks;
    // split-K (small feature maps at small batch: a handful of tiles, each with a K of thousands): blockIdx.y = split s runs
    // K units [s * kunits, (s + 1) * kunits) (conv_tap.hip: channel slices; conv.hip: K tiles) and stores its raw float32
    // accumulators to part[s][pixel][cout_pad]; splitk_reduce_kernel (aux.hip) sums them and runs the fused epilogue
    int ksplit, kunits, cout_pad;
    float *part;
    // split-K IN the launch (conv_tap.hip, 128 x 128 tile, ksplit == 2): both halves of a tile write their accumulators to
    // part[tile][half] (write-through), take a ticket from pair_cnt[tile], and the second arriver adds the other half and runs
    // the fused epilogue -- no reduce launch.  For launches of 129-256 tiles (13 x 13 / 19 x 19 maps at batch 8-32).
    int pair;
    int *pair_cnt;
    uint32_t part_bytes;
    int stream;                // conv_tap.hip: run the persistent (stream) form where it applies
    int fast_epi;              // conv_common.h: conv_epilogue_fast applies (set by the launchers: conv_fast_epilogue_ok)
    // back-to-back 1x1 (conv_common.h: conv_epilogue_fused_1x1): the 1x1 conv behind this one, computed by the same workgroups
    int fuse2;                 // 1: w2 .. are set and the launch runs the fused instantiation
    const void *w2;            // packed weights of the 1x1 ([128 rows][wrow2_bytes], natural filter order), 64 filters x 128 channels
    const float *b2;           // its folded bias
    void *out2;                // its output view (fp16, 16-byte aligned strides)
    uint32_t w2_bytes, wrow2_bytes, out2_bytes;
    int out2_ld, leaky2;
    long long out2_img_stride;
    int dbg;                   // experiment flags (YOLO_CONV_DBG): 1 skip steady-state DMA, 2 skip MFMA phase
    int f32_emu;               // conv.hip, float32 nets: this launch multiplies as nine bf16 products (conv_f32_emu_rule)
    int qW, qHW, Mq;           // conv_tap.hip: padded-linear pixel grid, row stride W+1, image stride (H+1)(W+1), total
    int q_stride;              // conv_tap.hip MODE 1: first position of tile m = m * q_stride (positions per tile; qHW for the image-aligned tile)
    int t2_shift;              // conv_tap.hip MODE 2: log2 of the positions per 2-D tile (8: 16 x 16, 7: 8 x 16)
    float *obj_out;            // head convs (staged float32 epilogue): compact objectness logits [B][obj_rows] or null
    int obj_width, obj_rows, obj_row0, obj_na;     // 5 + classes; rows per image; first row of this scale; anchors per cell
    float obj_min;             // head convs inside yolo_net_detect: rows whose objectness logit is below this are not written (conv_common.h); -inf: all rows
    unsigned long long *trace; // conv_tap.hip: per-block phase timestamps (YOLO_CONV_TRACE experiment) or null
    FastDiv dHoWo, dWo, dqHW, dqW, dtiles_n, dtpt;   // set by the launchers (conv_set_divisors); dtpt: K stages per tap
};
// fp16 output through the normal index map, 16-byte aligned views below 2 GiB, whole 16-cout groups, no split-K: the lean epilogue
inline bool conv_fast_epilogue_ok(const ConvParams &p) {
    return !p.f32 && !p.out_f32 && p.outmode == OUT_NORMAL && p.vec_out && p.out_bytes && p.Cout % 16 == 0 && p.ksplit <= 1 &&
           (!p.has_res || (p.vec_res && p.res_bytes));
}
inline void conv_set_divisors(ConvParams &p, int stages_per_tap) {
    p.dtpt = make_fastdiv((uint32_t)(stages_per_tap > 0 ? stages_per_tap : 1));
    p.dHoWo = make_fastdiv((uint32_t)p.HoWo);
    p.dWo = make_fastdiv((uint32_t)p.Wo);
    p.dqHW = make_fastdiv((uint32_t)(p.qHW > 0 ? p.qHW : 1));
    p.dqW = make_fastdiv((uint32_t)(p.qW > 0 ? p.qW : 1));
    p.dtiles_n = make_fastdiv((uint32_t)(p.n_tiles_n > 0 ? p.n_tiles_n : 1));
}

struct PrepParams {            // float32 NHWC [B,H,W,C] -> T NHWC [B,H,W,Cpad], zero fill
    const float *in;
    void *out;
    long long pixels;
    int C, Cpad;
};

struct FirstParams {           // first layer: 3x3/1 conv on the float32 NHWC3 input, Cout 16|32
    const float *in;           // [B,H,W,3] float32 (the caller's tensor)
    const float *wgt;          // [27][Cout] float32, BN folded (rounded through fp16 for fp16 nets)
    const float *bias;         // [Cout]
    void *out;                 // T NHWC view
    int H, W, Cout, out_ld, leaky, round_half;
    long long out_img_stride;
    long long total;           // B*H*W output pixels (< 2^31)
    FastDiv dW, dH, dHW;       // set by launch_first
    int pool;                  // 1: the 2x2/2 max-pool behind the conv is fused; `out` is the POOLED tensor [B,H/2,W/2,Cout]
    int xblocks;               // pool: 128-wide x blocks per row (set by launch_first); MFMA form: tiles per tile row
    FastDiv dXB, dHp;
    int tiles_y, n_tiles;      // MFMA form (first_pool_mfma_kernel): 8 x 16 pooled-output tiles per image column / in all
};

struct StemParams {            // stem.hip: fused conv 3x3/1 3->32 + conv 3x3/2 32->64 (both BN + leaky), fp16 nets
    const float *in;           // [B,H,W,3] float32 (the caller's tensor)
    const float *w1;           // first layer [27][32] float32 (K_FIRST packing)
    const float *b1;           // [32]
    const void *w2;            // second layer, K_CONV packing: [Cout_pad][wrow2 bytes] fp16, K = (kh, kw, 32 cin)
    const float *b2;           // [Cout_pad]
    void *out;                 // fp16 NHWC view [B,H/2,W/2,64]
    const void *w3;            // optional third layer (1x1 64->32, BN + leaky) on the same pixels: [Cout_pad][128 B] fp16, or null
    const float *b3;
    void *out3;                // fp16 NHWC view [B,H/2,W/2,32]
    int out3_ld;
    long long out3_img_stride;
    uint32_t w2_bytes, wrow2;
    int H, W, Ho, Wo, out_ld;
    long long in_img_stride, out_img_stride;
    int tiles_x, tiles_y, n_tiles;   // set by launch_stem
    FastDiv dtx, dty;
};

struct ResizeParams {          // uint8 HWC3 image -> float32 [dst_h][dst_w][3] in [0,1] (aux.hip: resize_u8_kernel)
    const unsigned char *src;
    float *dst;
    int src_h, src_w, src_row_bytes, dst_h, dst_w, swap_rb;
};

struct PoolParams {            // net/layers.py:70-81
    const void *in;
    void *out;
    int H, W, C, in_ld, Ho, Wo, out_ld, stride;
    long long in_img_stride, out_img_stride;
    long long total;           // B*Ho*Wo*(C/EPC) work items
};

struct EltParams {             // generic fallback: out[map(p)] = a[p] (+ b[p]); scalar, any view
    const void *a;
    const void *b;
    void *out;
    int H, W, C, a_ld, b_ld, out_ld, outmode, out_f32, a_f32;
    long long a_img_stride, b_img_stride, out_img_stride;
    long long total;           // B*H*W*C
};

struct ReduceParams {          // aux.hip: splitk_reduce_kernel -- sum of the K splits + the conv epilogue (conv_common.h semantics)
    const float *part;         // [ksplit][M][cout_pad]
    const float *bias;
    const void *res;           // residual (element pointer incl. coff) or null
    void *out;
    float *obj_out;            // head convs: compact objectness logits (see ConvParams)
    int obj_width, obj_rows, obj_row0, obj_na;
    int ksplit, M, Cout, cout_pad, HoWo, Wo;
    int out_ld, res_ld, leaky, outmode, out_f32, f32;
    long long out_img_stride, res_img_stride;
};

struct DecodeScale {
    int row0, h, w, na;
    double aw[YOLO_MAX_ANCHORS], ah[YOLO_MAX_ANCHORS];
};

struct DecodeParams {
    const float *logits;       // [B, rows, 5+C]
    const float *obj;          // optional compact copy of logits[..., 4] ([B, rows]): one coalesced read per row
    int version, n_classes, rows, n_scales;
    DecodeScale sc[YOLO_MAX_SCALES];
    float threshold;
    int cap;
    void *cand;                // Candidate[B][cap]
    int *cand_count;           // [B] counters, kCandCountStride ints apart (a 128-byte line each)
    long long total_rows;      // B*rows
};

struct Candidate {             // 40 bytes
    float x, y;
    double w, h;               // the reference computes w,h in float64 (anchors are np.float64)
    float prob;
    int cls;
    unsigned scan;             // scan index (cy, cw, anchor) across scales: stable-sort tie break
    int pad_;
};

struct NmsParams {
    const Candidate *cand;
    const int *cand_count;
    int *reset_count;          // non-null: the kernel returns cand_count[b] to zero once it has read it (the next decode needs no memset launch)
    int cap, max_boxes, mode;
    double iou_threshold;
    yolo_box *boxes;
    int *counts;
    int *status;
    int *keep_idx;             // optional [B][max_boxes]: candidate index of each survivor
    unsigned char *scratch;    // cap > 4096: [B][scratch_stride] bytes of global working storage (nms_scratch_bytes)
    size_t scratch_stride;
};

// ---- launchers (kernels.hip / detect.hip) ------------------------------------------------
hipError_t launch_conv(const ConvParams &p, int dtype, int cfg, bool perchunk, hipStream_t s);
// conv_dma.hip: 8-wave LDS-DMA kernel for the heavy fp16 layers.  choose_dma_cfg returns 0 when the
// 4-wave kernel of conv.hip should run, else the tile id for launch_conv_dma.
int choose_dma_cfg(int M, int cout, int cin_chunks, int taps, int has_res, bool v1_ok, int stride, int W, bool tap_only = false);   // -1: no DMA tile and no 4-wave kernel fits
bool dma_cfg_valid(int cfg, int cout, int cin_chunks, bool v1_ok, int ksize, int stride, int W);
bool dma_cfg_is_tap(int cfg);
int dma_cfg_bkc(int cfg);
hipError_t launch_conv_dma(const ConvParams &p, int cfg, hipStream_t s);
hipError_t launch_conv_tap(const ConvParams &p, int variant, hipStream_t s);       // conv_tap.hip: 3x3/1 with tap reuse
bool conv_tap_stream_ok(const ConvParams &p, int variant);                         // the persistent form takes this launch
bool conv_tap_fits(int variant, int W);
bool conv_tap_is2d(int variant);
bool conv_tap_stride2(int variant);         // 3x3 / stride 2 over the input's parity planes (MODE 4)
bool conv_tap_image_aligned(int variant);   // a tile = one whole image of the padded-linear grid (tile stride (H+1)(W+1))
bool conv_tap_f32_ok(int variant);            // float32 instantiation usable (tiles with room for the second accumulator)
bool dma_cfg_f32_ok(int cfg);
const char *dma_cfg_name(int cfg);
// names exactly as rocprofv3's kernel trace prints them (yolo_kernel_info.symbol: joins bench.py's roofline to profiles/*.csv)
const char *dma_cfg_symbol(int cfg, bool f32, bool fast = false);      // fast: the lean-epilogue instantiation of a tap tile
const char *conv_tap_symbol(int variant, bool f32, bool fast = false);
std::string dma_cfg_symbol_for(int cfg, bool f32, const ConvParams &p);      // the kernel that runs THIS launch (stream form included)
const char *conv_tap_stream_symbol(int variant);
std::string conv_symbol(int dtype, int cfg, bool perchunk, bool f32_emu = false);
// float32 nets: does this launch of the 4-wave kernel run its products as nine bf16 products (yolo_net_options.f32_products)?
bool conv_f32_emu_rule(int f32_products, int dtype, const ConvParams &p, int cfg, bool perchunk, int ksplit);
std::string first_symbol(int dtype, int cout, bool pool);
std::string aux_symbol(int kind, int dtype, bool vec);
int dma_num_cfgs();
int dma_cfg_na(int cfg);
int dma_cfg_nb(int cfg);
bool dma_cfg_splitk_ok(int cfg);        // the kernel behind this tile id takes ConvParams.ksplit
hipError_t launch_splitk_reduce(const ReduceParams &p, hipStream_t s);
bool conv_tap_splitk_ok(int variant);
bool conv_tap_pair_ok(int variant, bool f32);
hipError_t launch_prep(const PrepParams &p, int dtype, hipStream_t s);
hipError_t launch_resize(const ResizeParams &p, hipStream_t s);
hipError_t launch_first(const FirstParams &p, int dtype, hipStream_t s);
hipError_t launch_stem(const StemParams &p, int batch, hipStream_t s, int max_grid = 512);     // stem.hip
hipError_t launch_pool(const PoolParams &p, int dtype, hipStream_t s);
hipError_t launch_eltwise(const EltParams &p, int dtype, hipStream_t s);
hipError_t launch_decode(const DecodeParams &p, int batch, hipStream_t s, bool zero_counts = true);
hipError_t launch_nms(const NmsParams &p, int batch, hipStream_t s);
size_t nms_lds_bytes(int cap);
size_t nms_scratch_bytes(int cap);

// ---- plan -----------------------------------------------------------------------------------
struct Kernel {
    int kind = 0;
    int layer = -1;            // reference layer index this kernel materialises
    int src_layer = -1;        // conv: the conv layer whose weights it uses
    View in, in2, out;
    // conv
    int ksize = 0, stride = 0, cout = 0, cin = 0, cin_s = 0, leaky = 0, outmode = 0, has_res = 0;
    int cfg = 0, perchunk = 0, cpt = 0, ktiles = 0;
    int tile = -1;             // conv_dma tile id chosen by yolo_net_autotune (-1: heuristic)
    int head = 0;              // conv writes float32 head logits into the reference-layout output
    int pool_fused = 0;        // K_FIRST: the max-pool layer behind it is taken in the same kernel
    int stem = 0;              // 1: first-layer kernel fused away into the next conv; 2: this conv runs as stem.hip with it;
                               // 3: this 1x1 conv is computed by the stem kernel in front of it (no launch)
    int fuse2_next = 0;        // the NEXT kernel is a 1x1 128 -> 64 conv on this conv's output that the fused instantiation of this launch
                               // can compute (conv_common.h: conv_epilogue_fused_1x1); whether it does is decided per launch (api.cpp: conv_fuse2)
    int fuse2_prev = 0;        // ... and the mark on that 1x1: skipped when the conv in front of it has computed it
    int side = 0;              // > 0: member of branch tail `side` (plan.cpp: side_chains): a run of kernels ending in a head conv whose
                               // results nothing else reads -- may run on a second stream beside the kernels that follow it in the list
    size_t w_off = 0, b_off = 0, w_bytes = 0;   // inside the device weight blob
    size_t w_src = 0;                           // first float of this conv in the Darknet stream
    int batch_norm = 0;
    // pool
    int pool_stride = 0;
    std::string note;
};

struct LayerInfo {
    yolo_layer_desc d;
    int H = 0, W = 0, C = 0;
    std::vector<int> consumers;
    int fused_into = -1;       // conv kernel (layer idx) that produces this layer's tensor
    View view;                 // where the layer's output lives (after alias resolution)
    bool materialised = false;
};

}  // namespace yolo

struct yolo_net {
    yolo_net_options opt;
    std::vector<yolo::LayerInfo> layers;
    std::vector<yolo::Kernel> kernels;
    std::vector<yolo::Buffer> buffers;
    yolo_head_desc head;
    size_t weight_count = 0;       // floats in the Darknet stream
    size_t weights_bytes = 0;
    size_t act_bytes = 0;          // activation part of the workspace
    size_t logits_off = 0, cand_off = 0, count_off = 0, nms_off = 0;   // nms_off: global NMS slabs (cand_capacity > 4096)
    size_t splitk_off = 0, splitk_bytes = 0;   // float32 partial-sum slabs of the split-K convs (small feature maps at small batch)
    size_t obj_off = 0, obj_bytes = 0;     // compact objectness logits [max_batch][rows] written by the head convs for the decode
    bool obj_valid = false;                // ... and whether the last forward filled all of it
    int cand_clean = 0;                    // how many candidate counters, from the first, are known to be zero (the last detect's NMS returned them): no memset launch in front of a decode of at most that batch
    int side_chains = 0;                   // number of branch tails (Kernel.side ids 1..side_chains)
    std::vector<hipStream_t> branch;       // one stream per part for the branch tails, created at first use ...
    std::vector<hipEvent_t> e_bfork, e_bjoin;      // ... with fork events (4 per part) and one join event per part
    std::vector<signed char> side_ok;      // per batch: may the branch tails run beside the main chain (-1 unknown; no split-K launch in the pass)
    float obj_min_logit = -__builtin_inff();      // inside yolo_net_detect: objectness logit below which a row can never be a candidate (ConvParams.obj_min)
    std::vector<hipStream_t> side;         // multi-stream forward (YOLO_STREAMS=N): internal streams + fork/join events
    hipEvent_t e_fork = nullptr;
    std::vector<hipEvent_t> e_join;
    int arenas = 1;                        // activation arenas (2: one per half batch)
    // streams = 0 picked two parts by rule: every arena is then planned for the FULL batch, so that the same net can also run one pass
    // on one stream, and `parts` (1 or 2) says what a forward does -- the rule's answer until yolo_net_tune_streams has timed both
    // on this device (two halves gain 3-4 % on some MI355X and lose 1-2 % on others: profiles/r04_ablation.md section 7)
    bool arena_full = false;
    int parts = 1;                         // parts a full batch currently runs as (== arenas unless arena_full)
    bool parts_tuned = false;
    size_t arena_bytes = 0;
    bool halves = false;                   // the current forward runs as two concurrent half-batch passes
    size_t workspace_bytes = 0;
    size_t out_count = 0;          // floats per image of the head output
    double flops_per_image = 0;
    int esize = 2, epc = 8;
    // bound device memory
    unsigned char *dev_weights = nullptr;
    unsigned char *dev_ws = nullptr;
    size_t dev_ws_bytes = 0;
    bool weights_loaded = false;
};

namespace yolo {
int plan_network(yolo_net *net, const yolo_layer_desc *layers, int n, std::string &err);
int pack_weights(const yolo_net *net, const float *host, size_t n, std::vector<unsigned char> &blob, std::string &err);
std::string describe(const yolo_net *net);
void set_error(const std::string &s);
}  // namespace yolo
