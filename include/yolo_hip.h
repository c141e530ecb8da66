/*
 * yolo_hip.h -- C ABI of libyolo_hip.so: MI355X (gfx950) native YOLO v2/v3 TEST-mode hot path.
 *
 * Plain C: pointers, sizes and PODs only.  No torch / C++ types cross this boundary.
 * All device pointers are ordinary HIP device pointers (the Python host passes
 * torch-ROCm tensor .data_ptr() values); `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Every entry returns 0 on success or a YOLO_ERR_* code;
 * yolo_last_error() gives the message.  A yolo_net is used by one host thread at a time.
 * Unless stated otherwise work is ENQUEUED on `stream` and the caller synchronises.
 *
 * What each entry replaces in the reference (wns349/tensorflow-yolo):
 *   yolo_net_create         net/v2.py:11-60 create_full_network, net/v3.py:9-94 create_network
 *                           (the TF graph build: here a layer list -> fused kernel plan)
 *   yolo_net_load_weights   net/base.py:26-46 load_weights + net/v2.py:63-79 / net/v3.py:98-106
 *   yolo_net_forward        net/yolo.py:83  sess.run(net[-1].out, {net[0].out: x_batch})
 *   yolo_decode_nms         net/v2.py:83-119 / net/v3.py:140-151 find_bounding_boxes
 *                           (+ net/base.py:195-209 non_maximum_suppression)
 *   yolo_net_detect         net/yolo.py:83-86 (forward + find_bounding_boxes in one enqueue)
 *   yolo_nms_host           net/base.py:195-209 non_maximum_suppression on a host box list
 *   yolo_preprocess_resize  net/base.py:115-155 preprocess_image (resize + colour order + /255) on the device
 */
#ifndef YOLO_HIP_H
#define YOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YOLO_HIP_ABI_VERSION 5      /* 2: yolo_kernel_info.symbol; 3: yolo_net_num_streams, streams = 0 is the library's rule; 4: yolo_net_tune_streams; 5: yolo_net_set_streams */

enum yolo_status {
    YOLO_OK = 0,
    YOLO_ERR_ARG = 1,       /* bad argument                                        */
    YOLO_ERR_PLAN = 2,      /* layer list cannot be planned (unsupported topology) */
    YOLO_ERR_HIP = 3,       /* a HIP runtime call failed                           */
    YOLO_ERR_WEIGHTS = 4,   /* weight stream length != what the layer list needs   */
    YOLO_ERR_STATE = 5,     /* call order (weights / workspace not bound)          */
    YOLO_ERR_OVERFLOW = 6   /* more candidates than the configured capacity        */
};

/* Layer vocabulary == the reference's net/layers.py classes. */
enum yolo_op {
    YOLO_OP_INPUT = 0,      /* layers.py:106-109 input_layer     (h, w, c)                       */
    YOLO_OP_CONV = 1,       /* layers.py:17-67   conv2d_bn_act   (filters,ksize,stride,bn,leaky) */
    YOLO_OP_MAXPOOL = 2,    /* layers.py:70-81   max_pool2d      (ksize, stride)                 */
    YOLO_OP_ROUTE = 3,      /* layers.py:84-87   route           (src[0..n_src))                 */
    YOLO_OP_REORG = 4,      /* layers.py:90-97   reorg           (stride)                        */
    YOLO_OP_SHORTCUT = 5,   /* layers.py:100-103 shortcut        (src[0] = prev, src[1] = skip)  */
    YOLO_OP_UPSAMPLE = 6,   /* layers.py:112-116 upsample        (stride)                        */
    YOLO_OP_YOLO = 7,       /* layers.py:126-134 yolo_layer      (anchors in grid units)         */
    YOLO_OP_DETECTION = 8   /* layers.py:119-123 detection_layer (src = yolo layers)             */
};

enum yolo_dtype { YOLO_DTYPE_F32 = 0, YOLO_DTYPE_F16 = 1 };

enum yolo_nms_mode {
    YOLO_NMS_AGNOSTIC = 0,  /* the reference: class_idx never consulted (base.py:195-209) */
    YOLO_NMS_PER_CLASS = 1  /* opt-in: only boxes of the same class suppress each other   */
};

#define YOLO_MAX_SRC 4
#define YOLO_MAX_ANCHORS 8
#define YOLO_MAX_SCALES 4

/* One entry per element of the reference's `layers` list (index 0 = input layer). */
typedef struct yolo_layer_desc {
    int32_t op;                         /* enum yolo_op                                  */
    int32_t n_src;
    int32_t src[YOLO_MAX_SRC];          /* indices into the layer list                   */
    int32_t filters, ksize, stride;     /* conv / maxpool / reorg / upsample             */
    int32_t batch_norm;                 /* conv: 1 -> beta,gamma,mean,var,kernel stream  */
    int32_t leaky;                      /* conv: 1 -> leaky 0.1, 0 -> linear             */
    int32_t h, w, c;                    /* input layer                                   */
    int32_t n_anchors;                  /* yolo layer                                    */
    double anchors[2 * YOLO_MAX_ANCHORS];/* yolo layer: (w,h) pairs in GRID units (float64 like layers.py:131) */
} yolo_layer_desc;

typedef struct yolo_net_options {
    int32_t dtype;          /* enum yolo_dtype: storage/operand type (accumulation is f32)     */
    int32_t max_batch;      /* buffers are planned for this many images                        */
    int32_t keep_all;       /* 1: no activation-buffer reuse, so yolo_net_read_layer works     */
    int32_t cand_capacity;  /* candidates per image the decode stage can hold (0 -> 4096; up to
                             * 4096 sort + NMS run in LDS, up to 65536 on global-memory slabs; an image
                             * with at most 512 candidates always takes the single-wave LDS path)    */
    int32_t max_boxes;      /* records per image written by detect / decode_nms (0 -> 256)     */
    int32_t streams;        /* 1: one pass on the caller's stream; 2..4: the batch runs as that many independent
                             * parts on the caller's + internal streams (overlaps the kernels' tails and launch
                             * boundaries); 0: the library's rule (two parts for fp16 nets of >= 40 conv launches
                             * whose half batch is >= 2.5 M input pixels, e.g. YOLOv3-608 at batch >= 16, else one;
                             * the environment variable YOLO_STREAMS overrides the rule; yolo_net_tune_streams() re-measures
                             * the rule's "two" on the device).  Results do not depend on it up to fp16 summation order.
                             * yolo_net_num_streams() tells what a net runs with.                              */
    int32_t force_tile;     /* 0: per-layer tile choice (cost model / autotune); t + 1: run conv tile id t on every
                             * conv layer that accepts it (0 = 4-wave kernel, 1-7 and 14 LDS-DMA tiles, 8-13 and 15-17
                             * tap-reuse tiles): test and tuning hook, any value gives the same results up to summation order   */
    int32_t guard_bytes;    /* test hook (guard-band canaries): this many extra, never-used bytes behind every planned activation
                             * tensor (rounded into the tensor's 4 KiB-aligned region); with keep_all = 1 no two tensors share bytes, so a
                             * pattern-filled workspace shows any kernel that writes outside its tensor (yolo_net_workspace_regions);
                             * 0 in production                                                                                */
    int32_t f32_products;   /* float32 nets (ABI 5): how the long-K convs of the 4-wave kernel multiply.  0: the library's rule -- whole-K
                             * launches with K >= 4608 and >= 256 workgroups (tiny-YOLOv2's 13 x 13 512 -> 1024 and 1024 -> 1024 layers) run
                             * every float32 product as NINE bf16 x bf16 products on the bf16 matrix cores (a float32 value is exactly the
                             * sum of three bf16 values; each partial product is exact, the sums accumulate in float32: not narrower than
                             * tf.layers.conv2d's float32, net/layers.py:31-39; 16/9 of the float32 matrix rate); 1: native float32 MFMA
                             * everywhere; 2: nine bf16 products wherever the kernel applies.  Ignored by fp16 nets.                   */
} yolo_net_options;

/* Result record; field names follow net/base.py:257-272 BoundingBox. */
typedef struct yolo_box {
    float x, y, w, h;       /* centre / size, normalised to the image */
    float prob;
    int32_t class_idx;
} yolo_box;

/* Head geometry for the standalone decode (the reference reads the same from
 * net[-1].yolos[i].{h,w,b,anchors}, net/v3.py:145-149; v2 has one scale). */
typedef struct yolo_head_desc {
    int32_t version;                    /* 2: p = sigmoid(obj)*max softmax(cls); 3: p = sigmoid(obj) */
    int32_t n_classes;
    int32_t n_scales;
    int32_t h[YOLO_MAX_SCALES], w[YOLO_MAX_SCALES], n_anchors[YOLO_MAX_SCALES];
    double anchors[YOLO_MAX_SCALES][2 * YOLO_MAX_ANCHORS];  /* grid units, (w,h) pairs */
} yolo_head_desc;

typedef struct yolo_net yolo_net;

int yolo_hip_abi_version(void);
const char *yolo_last_error(void);                  /* thread-local message of the last failure */

/* ---- network object ------------------------------------------------------------------ */
int yolo_net_create(const yolo_layer_desc *layers, int n_layers, const yolo_net_options *opt, yolo_net **out);
void yolo_net_destroy(yolo_net *net);

size_t yolo_net_weight_count(const yolo_net *net);      /* float32 values the Darknet stream must hold  */
size_t yolo_net_weights_bytes(const yolo_net *net);     /* device bytes for the packed (BN-folded) weights */
size_t yolo_net_workspace_bytes(const yolo_net *net);   /* device bytes for activations + decode scratch */
size_t yolo_net_output_count(const yolo_net *net);      /* float32 values per image of the head output  */
double yolo_net_flops_per_image(const yolo_net *net);   /* sum 2*Ho*Wo*Cout*k*k*Cin over convs          */
int yolo_net_head_desc(const yolo_net *net, yolo_head_desc *out);
/* nets whose last layer is a plain conv (YOLOv2: the reference keeps anchors outside the graph,
 * net/v2.py:83-85) get their head geometry from the caller before detect() */
int yolo_net_set_head(yolo_net *net, const yolo_head_desc *head);
int yolo_net_num_kernels(const yolo_net *net);
int yolo_net_num_streams(const yolo_net *net);      /* parts / HIP streams a full batch currently runs as (yolo_net_options.streams) */
/* Where streams = 0 picked two halves by rule, both activation arenas hold a FULL batch and this call (optional, once, with a batch
 * above max_batch / 2; synchronous, 30 forward passes) times one pass against two halves on THIS device and keeps two halves where they win by 1.5 %: the same
 * build gains 3-4 % from two halves on one MI355X and loses 1-2 % on another (power-limited clocks differ from board to board).
 * No-op for nets whose streams were given explicitly or whose rule says one.  The Python engine calls it at its first full batch. */
int yolo_net_tune_streams(yolo_net *net, const float *in_dev, int batch, void *stream);
/* The same choice made by the caller instead of a measurement: 1 = one pass, 2 = two halves, for a net whose streams = 0 rule planned both
 * (else only the value it already runs with is accepted: YOLO_ERR_STATE otherwise).  For processes that must all run the SAME plan -- the ranks
 * of a sharded batch (net/dist.py broadcasts rank 0's measurement: fp16 sums of one pass and of two halves differ in order, and the step time of
 * the job is the slowest rank's).  Not a reference call site: net/yolo.py:65-67 is one process, one session. */
int yolo_net_set_streams(yolo_net *net, int parts);
/* Diagnostic (ABI 5; tests/test_gpu_ops.py::test_no_kernel_writes_outside_its_tensor): the regions of the workspace the plan laid out -- every
 * activation tensor of every arena, the head logits of detect(), candidate lists, counters, NMS scratch, the compact objectness array, the
 * split-K slabs.  Bytes [offset, offset + used_bytes) are the region's payload; [offset + used_bytes, offset + region_bytes) belong to it but are
 * never written by any kernel.  Fills at most `cap` records, returns the number of regions.  No reference call site (net/yolo.py holds no buffers). */
typedef struct yolo_ws_region {
    char name[32];
    uint64_t offset, used_bytes, region_bytes;
} yolo_ws_region;
int yolo_net_workspace_regions(const yolo_net *net, yolo_ws_region *out, int cap);
/* human-readable plan (kernels, fusions, buffers); returns bytes needed incl. NUL */
size_t yolo_net_describe(const yolo_net *net, char *buf, size_t cap);

/* host_weights: the float32 body of a Darknet .weights file (header stripped), n values, in
 * layer-list order.  dev_weights: caller-owned device memory of yolo_net_weights_bytes().
 * Folds BN in fp32/fp64 on the host, repacks to the kernel layout, copies H2D (synchronous). */
int yolo_net_load_weights(yolo_net *net, const float *host_weights, size_t n, void *dev_weights, size_t dev_bytes);
int yolo_net_bind_workspace(yolo_net *net, void *dev_workspace, size_t dev_bytes);

/* in_dev: float32 NHWC [batch,h,w,c] in [0,1] RGB (what net/base.py:115-155 produces, as f32).
 * out_dev: float32, reference layout -- v2 [B,h,w,A*(5+C)], v3 [B,sum(h*w*3),5+C] coarse->fine. */
int yolo_net_forward(yolo_net *net, const float *in_dev, int batch, float *out_dev, void *stream);

/* Optional: time every valid tile configuration of each heavy conv on the device (synchronous, a few
 * hundred launches) and keep the fastest per layer for later forward/detect calls at this batch. */
int yolo_net_autotune(yolo_net *net, const float *in_dev, int batch, void *stream);

/* forward + decode + NMS.  boxes_dev: [batch][max_boxes] yolo_box, counts_dev: [batch] int32
 * (number of valid records, descending prob, stable); status_dev: [batch] int32
 * (0 ok, 1 candidate overflow, 2 more survivors than max_boxes: list truncated).
 * threshold is compared in float32 (`p < threshold`, net/v2.py:107), iou_threshold in float64
 * (`iou >= iou_threshold`, net/base.py:204), as NumPy does in the reference. */
int yolo_net_detect(yolo_net *net, const float *in_dev, int batch, double threshold, double iou_threshold,
                    int nms_mode, yolo_box *boxes_dev, int32_t *counts_dev, int32_t *status_dev, void *stream);

/* Per-kernel facts for roofline accounting (bench.py): algorithmic work of ONE image. */
typedef struct yolo_kernel_info {
    int32_t kind;               /* 0 prep, 1 conv, 2 maxpool, 3 eltwise                         */
    int32_t layer;              /* reference layer index the kernel materialises                 */
    int32_t variant;            /* conv: cout-tile config (0 N128, 1 N64, 2 N32) + 4*perchunk    */
    int32_t ksize, stride, cin, cout, out_h, out_w;
    double flops;               /* conv: 2*Ho*Wo*Cout*k*k*Cin, else 0                            */
    double bytes;               /* input + output (+ residual) elements * element size           */
    double weight_bytes;        /* packed weights + bias read once per launch (not per image)    */
    char name[64];              /* readable label of the kernel family, e.g. "conv_igemm_dma<f16,128x256,tap9,x2>" */
    char symbol[160];           /* the kernel's name exactly as rocprofv3 --kernel-trace prints it, e.g.
                                 * "void yolo::conv3x3_tap_kernel<false, 2, 4, 4, 4, 26, 4, 1>(yolo::ConvParams)":
                                 * joins roofline.kernel_symbol of bench.py to the kernel_stats CSVs under profiles/; "" = no launch */
} yolo_kernel_info;
int yolo_net_kernel_info(const yolo_net *net, int kernel, yolo_kernel_info *out);

/* yolo_net_forward with every kernel bracketed by hipEvents recorded on `stream`; synchronises and
 * writes the device time of each kernel in milliseconds to ms_host[yolo_net_num_kernels()].
 * Measurement aid only (the events add bubbles): never used for throughput numbers. */
int yolo_net_forward_timed(yolo_net *net, const float *in_dev, int batch, float *out_dev, void *stream, float *ms_host);

/* debug / parity: copy one layer's output to host as dense float32 NHWC (needs keep_all; synchronous) */
int yolo_net_read_layer(yolo_net *net, int layer, int batch, float *host_out, size_t n);

/* ---- standalone decode + NMS (drop-in for find_bounding_boxes) --------------------- */
size_t yolo_decode_scratch_bytes(const yolo_head_desc *head, int batch, int cand_capacity);
int yolo_decode_nms(const yolo_head_desc *head, const float *logits_dev, int batch, double threshold,
                    double iou_threshold, int nms_mode, int cand_capacity, int max_boxes, void *scratch_dev,
                    size_t scratch_bytes, yolo_box *boxes_dev, int32_t *counts_dev, int32_t *status_dev,
                    void *stream);

/* Image preprocessing of the TEST loop, replaces net/base.py:115-155 (cv2.resize INTER_LINEAR stretch to the network
 * input, optional BGR<->RGB swap, / 255.): src_dev is a decoded uint8 HWC image with 3 channels on the device
 * (src_row_bytes >= 3*src_w), dst_dev receives float32 [dst_h][dst_w][3] in [0,1].  Bit-exact with OpenCV's published
 * 8-bit INTER_LINEAR algorithm as restated in oracle/preprocess_ref.py (OpenCV itself is not available: unpinned).
 * Enqueued on `stream`. */
int yolo_preprocess_resize(const uint8_t *src_dev, int src_h, int src_w, int src_row_bytes, float *dst_dev, int dst_h, int dst_w,
                           int swap_rb, void *stream);

/* NMS of a HOST list (x,y,w,h as double, prob float, class int; scan order = index).  Synchronous;
 * allocates its own scratch.  keep_idx receives the indices of survivors in output order.
 * x and y are rounded to float32 before the IoU arithmetic: that is the type the reference's decode gives them
 * (net/v2.py:112-113, net/v3.py:129-130 under NumPy 2) and what every golden vector holds; a caller passing
 * float64 centres that are not float32 values gets the float32-rounded result (w, h stay float64). */
int yolo_nms_host(const double *xywh, const float *prob, const int32_t *class_idx, int n, double iou_threshold,
                  int nms_mode, int32_t *keep_idx, int32_t *n_keep);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HIP_H */
