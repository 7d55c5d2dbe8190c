/*
 * mi355q.h -- C ABI of libmi355q.so, the MI355X (gfx950) quantized-decode
 * kernel library.  Plain pointers and sizes only; no torch types.
 *
 * Every device pointer is memory owned by the caller (torch's caching
 * allocator in the vLLM integration).  No entry point allocates or frees device
 * memory, synchronises the host, or touches any stream other than the one
 * passed in, so all of them can be captured into a HIP graph.  (Two kernels keep
 * arrival counters in a small static device array of the library, one bank per stream:
 * see mi355q_paged_attention, mi355q_argmax and mi355q_reset_library_state.)
 *
 * Return value: 0 on success, a negative MI355Q_E* code otherwise;
 * mi355q_last_error() returns a thread-local message for the last failure
 * (the Python side turns it into RuntimeError, as TORCH_CHECK does in the
 * reference).
 *
 * "Replaces" comments cite the reference interface (file:line, relative to the
 * vLLM fork's tree) that each entry point stands in for.
 */
#ifndef MI355Q_H
#define MI355Q_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mi355q_stream_t; /* hipStream_t */

enum {
  MI355Q_OK = 0,
  MI355Q_EINVAL = -1,   /* bad shape / alignment / enum */
  MI355Q_ENOSPACE = -2, /* workspace too small */
  MI355Q_EHIP = -3      /* HIP runtime error at launch */
};

enum { MI355Q_BF16 = 0, MI355Q_FP16 = 1, MI355Q_FP32 = 2 };

/* checkpoint layouts of 4-bit weights accepted by the repack */
enum {
  MI355Q_W_GPTQ = 0, /* int32 [K/8, N], 8 consecutive k per word (auto_gptq.py:381-445) */
  MI355Q_W_NSEQ = 1, /* int32 [K, N/8], 8 consecutive n per word (triton_w4a16.py:368-389) */
  MI355Q_W_AWQ = 2   /* int32 [K, N/8], AWQ nibble order (auto_awq.py:478-515) */
};
/* OR-ed into src_layout: the packed matrix interleaves the two halves of the checkpoint's
 * output channels (packed column 2i = column i, 2i+1 = column N/2 + i), i.e. gate_i and up_i
 * of a merged gate_up projection (linear.py:661-1020) side by side.  The caller permutes the
 * scales the same way; zc comes out permuted. */
enum { MI355Q_W_PAIR_HALVES = 0x100 };

/* zero-point conventions accepted by the repack */
enum {
  MI355Q_ZP_CONST = 0,  /* symmetric uint4b8: z = 8, qzeros ignored (triton_w4a16.py:440-444) */
  MI355Q_ZP_NSEQ = 1,   /* int32 [K/G, N/8] sequential nibbles, value z */
  MI355Q_ZP_AWQ = 2,    /* int32 [K/G, N/8] AWQ nibble order, value z */
  MI355Q_ZP_GPTQ_V1 = 3 /* int32 [K/G, N/8] sequential, stores z-1 (q_gemm.cu:202) */
};

const char* mi355q_last_error(void);
int mi355q_version(void);
/*
 * The library reads its MI355Q_* environment knobs (route selection for tests and experiments,
 * e.g. MI355Q_PA_REDUCE_KERNEL, MI355Q_ARGMAX_NARROW, MI355Q_SPLITK) once per process, at first use.
 * This sets (value != NULL) or clears (value == NULL) one of them for the rest of the process.
 * No reference counterpart (the reference reads vllm.envs at import).
 */
int mi355q_set_option(const char* name, const char* value);
/*
 * Zeroes the arrival counters and result slots that mi355q_paged_attention (and the two attention
 * entry points beside it) and mi355q_argmax keep in device memory of the library.  Every call leaves
 * them zero, so this is only needed after a launch was aborted half-way (device fault, process-level
 * recovery).  No reference counterpart.  Launches a small kernel on `stream`.
 */
int mi355q_reset_library_state(mi355q_stream_t stream);

/* ------------------------------------------------------------------ */
/* W4A16 dequant-GEMM                                                  */
/* ------------------------------------------------------------------ */

/*
 * Strip width (64, 32 or 16 output channels) of the tiled layout for a [K, N]
 * layer: the narrowest strips that still give each of the 256 CUs about one
 * strip.  The same value must be passed to the repack and to every GEMM.
 */
int mi355q_w4a16_choose_tile(int K, int N, int G);
/*
 * Strip width for layers that serve decode: any of 16, 32, 64, 112 -- chosen so that the layer
 * gives about one workgroup per CU (N = 28672 -> 112 channels x 256 strips).  Falls back to
 * mi355q_w4a16_choose_tile when the group size is not a multiple of 128 or the layer is too
 * narrow to occupy the chip without splitting K across workgroups.
 */
int mi355q_w4a16_choose_strip(int K, int N, int G);
/* The same for the experts of a routed layer, where a decode launch runs decode_slots (tokens x top-k) rows with a
 * workgroup per strip each: Mixtral's w2 (14336 -> 4096) at top-2 takes 32-wide strips, one round of 256 workgroups
 * instead of two on 16-wide ones.  No counterpart in the reference (its expert GEMM has one tile shape per dtype,
 * csrc/moe/moe_wna16.cu:16-350). */
int mi355q_moe_w4a16_choose_strip(int K, int N, int G, int decode_slots);

/* 2- and 3-bit GPTQ tensors -> the 4-bit checkpoint layout mi355q_w4a16_repack takes (device, load time).
 *   axis 0: src int32 [values * bits / 32, other], one bit string per column along K (qweight)
 *           -> dst int32 [values / 8, other], nibble j of word r = value 8 r + j
 *   axis 1: src int32 [other, values * bits / 32], one bit string per row along N (qzeros)
 *           -> dst int32 [other, values / 8]
 * Value i occupies bits [bits i, bits i + bits) of its string (32-bit words in order); `add` is added to
 * every value (symmetric weights: 8 - 2^(bits-1), so that the kernels' constant zero point 8 applies).
 * Replaces the 2- / 3-bit branches of gptq_shuffle + gptq_gemm
 * (csrc/libtorch_stable/quantization/gptq/q_gemm.cu:323-682, qdq_2.cuh, qdq_3.cuh): the widened
 * weights run on the W4A16 kernels. */
int mi355q_gptq_expand_bits(const void* src, void* dst, int values, int other, int bits, int add, int axis,
                            mi355q_stream_t stream);

/*
 * Load-time repack of a 4-bit checkpoint tensor into the kernel's tiled layout
 * (1 KiB tiles of `tile` output channels x 2048/tile input channels, tiles of one
 * `tile`-wide strip contiguous along K; within a tile lane l of a 64-lane wavefront
 * owns 16 bytes = the MFMA B-operand fragments it will feed, so a wavefront reads
 * one tile with a single coalesced 16-byte-per-lane load).
 * Replaces MPLinearKernel.process_weights_after_loading
 * (mixed_precision/MPLinearKernel.py:56-58; triton_w4a16.py:338-426;
 * exllama.py:60-136 + gptq_shuffle q_gemm.cu:1857-1868).
 *   qweight : device, layout per src_layout
 *   qzeros  : device or NULL (required unless zp_mode == MI355Q_ZP_CONST)
 *   packed  : device out, K*N/2 bytes
 *   zc      : device out, (K/G)*N 16-bit values of (bias + z) in act_dtype, strip-major like the
 *             scales (see mi355q_w4a16_gemm), or NULL when zp_mode == MI355Q_ZP_CONST
 * Requires K % 32 == 0, N % 64 == 0, G % 32 == 0 and K % G == 0 (G <= 0 means K).
 */
int mi355q_w4a16_repack(const int32_t* qweight, const int32_t* qzeros, void* packed, void* zc,
                        int K, int N, int G, int tile, int src_layout, int zp_mode, int act_dtype,
                        mi355q_stream_t stream);
/* Same arithmetic on host memory (used by load paths that repack before upload
 * and by the CPU-only tests). */
int mi355q_w4a16_repack_host(const int32_t* qweight, const int32_t* qzeros, void* packed, void* zc,
                             int K, int N, int G, int tile, int src_layout, int zp_mode,
                             int act_dtype);
/* Inverse of the weight repack: packed tiles -> int32 [K, N] values in [0,16). */
int mi355q_w4a16_unpack_host(const void* packed, int32_t* q, int K, int N, int tile);

size_t mi355q_w4a16_packed_bytes(int K, int N);
/*
 * Split-K workspace: a block of arrival counters followed by fp32 partial tiles.
 * The counters must be zero before the first GEMM that uses the workspace
 * (mi355q_workspace_init, once, outside any graph capture); every GEMM leaves
 * them zero again.  GEMMs that share one workspace must be ordered on one
 * stream.  The reference allocates its scratch inside the op instead
 * (q_gemm.cu:1835-1838).
 */
size_t mi355q_w4a16_workspace_bytes(int M, int N, int K, int G);
int mi355q_workspace_init(void* workspace, size_t workspace_bytes, mi355q_stream_t stream);

/*
 * C[M,N] = A[M,K] * ((q - z) * s)  (+ bias), fp32 accumulation, one rounding.
 * Replaces ops.gptq_gemm (vllm/_custom_ops.py:558-577; q_gemm.cu:1827-1855),
 * ops.awq_gemm (:525-536; awq/gemm_kernels.cu:472-536) and triton_w4a16_gemm
 * (mixed_precision/triton_w4a16.py:164-270).
 *   a      : [M, K] act_dtype, row stride lda elements (lda % 8 == 0)
 *   packed : output of mi355q_w4a16_repack
 *   scales : act_dtype, strip-major [N/tile][K/G][tile]: element (g, n) of the checkpoint's
 *            [K/G, N] tensor at ((n / tile) * (K/G) + g) * tile + n % tile, so that one
 *            workgroup's scales are one contiguous run (zc from the repack uses the same order)
 *   zc     : from the repack, or NULL for symmetric uint4b8
 *   bias   : [N] act_dtype or NULL
 *   c      : [M, N] act_dtype, contiguous
 *   tile      : strip width the weight was repacked with
 *   workspace : >= mi355q_w4a16_workspace_bytes(M,N,K,G) bytes (only touched when K is split
 *               across workgroups)
 *   split_k   : 0 = choose automatically
 */
int mi355q_w4a16_gemm(const void* a, int lda, const void* packed, const void* scales,
                      const void* zc, const void* bias, void* c, int M, int N, int K, int G,
                      int tile, int act_dtype, void* workspace, size_t workspace_bytes,
                      int split_k, mi355q_stream_t stream);

/*
 * Same GEMM with A formed on the fly as silu(x[:, :K]) * x[:, K:] from
 * x [M, 2K] (the gate_up projection output), i.e. SiluAndMul fused into the
 * down projection.  Replaces torch.ops._C.silu_and_mul
 * (activation_kernels.cu:98-155) + the GEMM above.
 */
int mi355q_w4a16_gemm_silu_mul(const void* x, int ldx, const void* packed, const void* scales,
                               const void* zc, const void* bias, void* c, int M, int N, int K,
                               int G, int tile, int act_dtype, void* workspace,
                               size_t workspace_bytes, int split_k, mi355q_stream_t stream);

/*
 * Same GEMM with the preceding fused_add_rms_norm folded into its prologue:
 *   z = round(x + residual_in)            (residual_in may be NULL: z = x)
 *   residual_out = z                      (must not alias x or residual_in)
 *   A = round(z * rsqrt(mean(z^2) + eps) * norm_weight)
 *   C = A * dequant(W) (+ bias)
 * Requires K = hidden size of the norm.  Replaces torch.ops._C.fused_add_rms_norm
 * (layernorm_kernels.cu:106-217) / rms_norm followed by the GEMM above.
 */
int mi355q_w4a16_gemm_add_rms_norm(const void* x, int ldx, const void* residual_in,
                                   void* residual_out, const void* norm_weight, float epsilon,
                                   const void* packed, const void* scales, const void* zc,
                                   const void* bias, void* c, int M, int N, int K, int G, int tile,
                                   int act_dtype, void* workspace, size_t workspace_bytes,
                                   int split_k, mi355q_stream_t stream);

/*
 * The same GEMM for many rows (prompt processing; worthwhile from 17 rows): tiled matrix-core
 * kernels that read the packed weight once per 64 or 128 rows (through an LDS ring fed by LDS-DMA
 * from 33 rows on) instead of expanding it to 16 bits first.
 * Same arithmetic as mi355q_w4a16_gemm (fp32 (q - z) * s, one rounding, bias after rounding).
 * Replaces the many-token routes of the reference: exllama reconstruct + hipBLAS
 * (q_gemm.cu:1509-1522), awq_dequantize + matmul (auto_awq.py:947-955) and the large-M tiles of
 * triton_w4a16_gemm (triton_w4a16.py:164-270).
 *   silu_pairs : 1 = weight repacked with MI355Q_W_PAIR_HALVES, write
 *                c[M, N/2] = silu(y[:, 2i]) * y[:, 2i+1] (no bias), as mi355q_w4a16_gate_up_silu
 *   workspace  : >= mi355q_w4a16_mm_workspace_bytes(M, N, K, G, tile) bytes of device memory (fp32
 *                partial tiles of layers whose K is sliced over workgroups; 0 for wide layers).
 *                Needs no initialisation; GEMMs sharing it must be ordered on one stream.
 * Requires G % 128 == 0 and N a multiple of 256 (448 for 112-wide strips):
 * mi355q_w4a16_mm_supported(N, K, G, tile) tells; callers fall back to
 * mi355q_w4a16_dequantize + a dense GEMM otherwise.
 * mi355q_w4a16_mm_min_rows(N, K, G, tile): the row count from which this entry is the faster one
 * for the weight (17; 13 on 112-wide strips, where the few-row GEMV entries take a second pass
 * over the weights) -- the reference draws the same line inside its dispatch heuristics
 * (gptq exllama: reconstruct above 50 rows, q_gemm.cu:1509; awq: 256 tokens, auto_awq.py:947).
 */
size_t mi355q_w4a16_mm_workspace_bytes(int M, int N, int K, int G, int tile);
int mi355q_w4a16_mm_supported(int N, int K, int G, int tile);
int mi355q_w4a16_mm_min_rows(int N, int K, int G, int tile);
/* 1 when mi355q_w4a16_mm serves M rows of this weight on the kernel that streams every strip once and needs no
 * K-slices (decode batches): a caller that would otherwise fuse the residual add and the norm behind the K-slice
 * merge (mi355q_w4a16_mm_add_rms_norm) is faster with this entry plus mi355q_fused_add_rms_norm on 16-wide strips.
 * Part of the same dispatch heuristics as above (q_gemm.cu:1509, auto_awq.py:947). */
int mi355q_w4a16_mm_streams(int M, int N, int K, int G, int tile);
int mi355q_w4a16_mm(const void* a, int lda, const void* packed, const void* scales, const void* zc,
                    const void* bias, void* c, int M, int N, int K, int G, int tile, int act_dtype,
                    int silu_pairs, void* workspace, size_t workspace_bytes, mi355q_stream_t stream);
/*
 * mi355q_w4a16_gemm_add_rms_norm whose input row is the combine of a routed layer: x = round(w0 y0 + w1 y1) in
 * fp32 in slot order, expert_out [1, 2, K], topk_weights fp32 [1, 2] -- the bits of mi355q_moe_sum -- so that the
 * decode step of a top-2 mixture-of-experts model needs no launch between the experts' down projection and the
 * next layer's qkv projection (mixtral.py:126-130 final_hidden_states -> :280-284 input_layernorm of the next
 * layer).  One token, two selected experts, a layer mi355q_..._supported accepts; other cases run mi355q_moe_sum
 * and mi355q_w4a16_gemm_add_rms_norm.
 */
int mi355q_w4a16_gemm_moe_sum_add_rms_norm_supported(int num_tokens, int topk, int N, int K, int G, int tile);
int mi355q_w4a16_gemm_moe_sum_add_rms_norm(const void* expert_out, const float* topk_weights, int topk,
                                           const void* residual_in, void* residual_out, const void* norm_weight,
                                           float epsilon, const void* packed, const void* scales, const void* zc,
                                           const void* bias, void* c, int num_tokens, int N, int K, int G, int tile,
                                           int act_dtype, mi355q_stream_t stream);
/* mi355q_w4a16_mm followed by mi355q_fused_add_rms_norm on its result (row-parallel projection ->
 * residual add -> RMSNorm, vllm/model_executor/models/llama.py:311-328): residual = round(c + residual),
 * out = RMSNorm(residual) * norm_weight.  When the GEMM is cut into K-slices, the slice sums, the
 * bias, the residual add and the norm run as one kernel over the fp32 partial tiles.  Same
 * workspace as mi355q_w4a16_mm. */
int mi355q_w4a16_mm_add_rms_norm(const void* a, int lda, const void* packed, const void* scales,
                                 const void* zc, const void* bias, void* residual,
                                 const void* norm_weight, float epsilon, void* out, int M, int N,
                                 int K, int G, int tile, int act_dtype, void* workspace,
                                 size_t workspace_bytes, mi355q_stream_t stream);
/* Whether mi355q_w4a16_mm_add_rms_norm serves this shape (17+ rows on the tiled kernel, 5..16 rows
 * when the K-sliced rows kernel is the route mi355q_w4a16_gemm would take), and the fp32 workspace
 * it then needs. */
int mi355q_w4a16_mm_add_rms_norm_supported(int M, int N, int K, int G, int tile, int has_zc);
size_t mi355q_w4a16_mm_add_rms_norm_workspace_bytes(int M, int N, int K, int G, int tile, int has_zc);
/* mi355q_w4a16_mm for the fused qkv projection of prompt tokens followed by mi355q_rope_and_cache on
 * its q / k / v column ranges (QKVParallelLinear -> rotary_emb -> cache update,
 * vllm/model_executor/models/llama.py:218-226); K-slices are merged by mi355q_splitk_rope_and_cache.
 * Same workspace as mi355q_w4a16_mm. */
int mi355q_w4a16_mm_rope_and_cache(const void* a, int lda, const void* packed, const void* scales,
                                   const void* zc, const void* bias, void* qkv, int M, int N, int K, int G,
                                   int tile, int act_dtype, void* workspace, size_t workspace_bytes,
                                   const int64_t* positions, int head_size, int num_heads, int num_kv_heads,
                                   const void* cos_sin_cache, int rot_dim, int cache_dtype, int is_neox,
                                   void* key_cache, void* value_cache, const int64_t* slot_mapping,
                                   int block_size, int64_t block_stride, int kv_fp8, const float* k_scale,
                                   const float* v_scale, mi355q_stream_t stream);

/*
 * gate_up projection with SiluAndMul in its epilogue, for weights repacked with
 * MI355Q_W_PAIR_HALVES: c[M, N/2] = silu(y[:, 2i]) * y[:, 2i+1] with y = the GEMM above
 * (rounded to act_dtype first, silu rounded before the multiply: the arithmetic of
 * torch.ops._C.silu_and_mul on the projection output, activation_kernels.cu:98-155).
 * residual_out != NULL additionally runs the add + RMSNorm prologue of
 * mi355q_w4a16_gemm_add_rms_norm; residual_out == NULL takes x as is.  The down projection
 * then reads N/2 activations per row instead of N.  Needs a layer the burst kernel serves
 * (group size a multiple of 128, N / tile >= 128); returns MI355Q_EINVAL otherwise.
 */
/* 1 when mi355q_w4a16_gate_up_silu can serve a [K, N] layer repacked with this tile width. */
int mi355q_w4a16_gate_up_silu_supported(int K, int N, int G, int tile);
int mi355q_w4a16_gate_up_silu(const void* x, int ldx, const void* residual_in, void* residual_out,
                              const void* norm_weight, float epsilon, const void* packed,
                              const void* scales, const void* zc, void* c, int M, int N, int K, int G,
                              int tile, int act_dtype, mi355q_stream_t stream);

/* W[K,N] (act_dtype) = (q - z) * s.  Replaces ops.awq_dequantize
 * (vllm/_custom_ops.py:491-505; awq/gemm_kernels.cu:414-464). */
int mi355q_w4a16_dequantize(const void* packed, const void* scales, const void* zc, void* w,
                            int K, int N, int G, int tile, int act_dtype, mi355q_stream_t stream);

/* ------------------------------------------------------------------ */
/* Mixture of experts, 4-bit experts (MoE WNA16)                       */
/* ------------------------------------------------------------------ */
/*
 * Router: probs = softmax(gating_output[t, :]) in fp32, the topk largest (ties: lowest expert
 * index) with their probabilities, optionally renormalised to sum to 1.  Replaces ops.topk_softmax
 * (vllm/_custom_ops.py:2385; csrc/libtorch_stable/moe/topk_softmax_kernels.cu).
 *   topk_weights fp32 [T, topk], topk_ids int32 [T, topk], gating_output [T, E] in `dtype`.
 */
int mi355q_topk_softmax(float* topk_weights, int32_t* topk_ids, const void* gating_output,
                        int num_tokens, int num_experts, int topk, int renormalize, int dtype,
                        mi355q_stream_t stream);
/*
 * The router of a decoder layer at decode sizes as one launch, one workgroup per token: the
 * post-attention fused_add_rms_norm (in place on input and residual, as mi355q_fused_add_rms_norm), the
 * 16-bit gate projection logits[t, e] = round(input[t, :] . router_weight[e, :]) and mi355q_topk_softmax
 * of the logits, each stage with the arithmetic of its stand-alone entry point (same bits).  Replaces the
 * sequence post_attention_layernorm -> gate -> fused_topk of vllm/model_executor/models/mixtral.py:284,
 * 144-146.  logits ([T, E] in `dtype`) may be NULL.  Up to 64 experts, hidden a multiple of 8 up to 8192
 * (mi355q_moe_router_topk_supported); other layers run the three entry points one after the other.
 */
int mi355q_moe_router_topk_supported(int num_experts, int hidden);
int mi355q_moe_router_topk(void* input, int64_t input_stride, void* residual, const void* norm_weight,
                           float epsilon, const void* router_weight, void* logits, float* topk_weights,
                           int32_t* topk_ids, int num_tokens, int num_experts, int hidden, int topk,
                           int renormalize, int dtype, mi355q_stream_t stream);
/*
 * Expert GEMM for decode: output row r ("slot" = one (token, selected expert) pair) is
 *   c[r, :] = x[r / x_row_div, :] * dequant(W[expert_ids[r]])      (fuse_silu_mul = 0)
 *   c[r, :] = (silu(x[r, :K]) * x[r, K:2K]) * dequant(W[expert_ids[r]])   (fuse_silu_mul = 1)
 *   c[r, :N/2] = silu(y[2i]) * y[2i+1], y = x[r / x_row_div, :] * dequant(W[...])   (fuse_silu_mul = 2:
 *                experts repacked with MI355Q_W_PAIR_HALVES, SiluAndMul in the epilogue as in
 *                mi355q_w4a16_gate_up_silu; needs a layer the one-workgroup-per-strip kernel serves,
 *                mi355q_w4a16_gate_up_silu_supported)
 * with every expert's weight repacked by mi355q_w4a16_repack (same tile width) and stored
 * back to back: packed [E][K*N/2 bytes], scales [E][K/G][N], zc [E][K/G][N] or NULL.
 * expert_ids[r] < 0 (expert held by another rank) zero-fills the row.  No token sorting /
 * padding pass is needed (the reference runs moe_align_block_size first).  Replaces
 * ops.moe_wna16_gemm (vllm/_custom_ops.py:2307-2342; csrc/moe/moe_wna16.cu:16-350) and the Triton
 * fused_moe_kernel_gptq_awq (fused_moe.py:64-296) for small token counts; x_row_div = topk for
 * the first GEMM (w13: every slot of a token reads the token's row), 1 for the second (w2).
 */
int mi355q_moe_w4a16_gemm(const void* x, int ldx, int x_row_div, const void* packed,
                          const void* scales, const void* zc, const int32_t* expert_ids, void* c,
                          int slots, int num_experts, int N, int K, int G, int tile, int act_dtype,
                          int fuse_silu_mul, mi355q_stream_t stream);
/*
 * Token sort for many-token batches.  expert_ids[slots]: the (local) expert of every (token, choice) slot,
 * negative for experts held by another rank.  Slots are grouped by expert; every expert's run is padded to
 * whole tiles of `block` slots:
 *   sorted_ids[max_tiles * block]  slot index per sorted row, `slots` for padding rows
 *   tile_expert[max_tiles]         expert of every tile, -1 past the last used tile
 *   num_tiles[1]                   number of used tiles
 *   slot_pos[slots]                sorted row of every slot (-1: expert not held here)
 * max_tiles >= ceil((slots + min(slots, E) * (block - 1)) / block).  The order of an expert's slots inside its
 * run is not defined (no result depends on it).  Replaces ops.moe_align_block_size
 * (vllm/_custom_ops.py:2237-2255; csrc/libtorch_stable/moe/moe_align_sum_kernels.cu:326-700; the
 * reference's expert_ids per block / num_tokens_post_pad are tile_expert / num_tiles * block here).
 */
int mi355q_moe_align_block_size(const int32_t* expert_ids, int slots, int num_experts, int block,
                                int32_t* sorted_ids, int32_t* tile_expert, int32_t* num_tiles, int32_t* slot_pos,
                                int max_tiles, mi355q_stream_t stream);
/* Row-tile height of mi355q_moe_w4a16_grouped_mm: the `block` to sort with. */
int mi355q_moe_grouped_block_rows(void);
size_t mi355q_moe_w4a16_grouped_mm_workspace_bytes(int rows, int N, int K, int G, int tile);
/*
 * Grouped expert GEMM over token-sorted rows: c[r, :] = a[r, :] @ dequant(expert tile_expert[r / block]) for the
 * rows of the first *num_tiles tiles (other rows are left untouched); a [rows, K] holds the gathered
 * activations in sorted order (padding rows: any finite values), rows = max_tiles * block.  Same weights,
 * arithmetic and rounding as mi355q_moe_w4a16_gemm; silu_pairs: SiluAndMul in the epilogue, c is [rows, N/2]
 * (experts repacked with MI355Q_W_PAIR_HALVES).  Replaces ops.moe_wna16_gemm for prompt batches
 * (vllm/_custom_ops.py:2307-2342; csrc/libtorch_stable/moe/moe_wna16.cu:16-350) on the sorted layout of
 * fused_experts_impl (fused_moe.py:1650-1850).
 */
int mi355q_moe_w4a16_grouped_mm(const void* a, int lda, const void* packed, const void* scales, const void* zc,
                                const int32_t* tile_expert, const int32_t* num_tiles, void* c, int rows,
                                int num_experts, int N, int K, int G, int tile, int act_dtype, int silu_pairs,
                                void* workspace, size_t workspace_bytes, mi355q_stream_t stream);
/*
 * out[t, :] = round(sum_j topk_weights[t, j] * expert_out[t * topk + j, :]), fp32 sum in
 * slot order (topk_weights NULL: plain sum).  Replaces ops.moe_sum (vllm/_custom_ops.py:2228;
 * moe_align_sum_kernels.cu) together with the routed-weight multiply of the expert GEMM
 * epilogue (fused_moe.py MUL_ROUTED_WEIGHT).
 */
int mi355q_moe_sum(void* out, const void* expert_out, const float* topk_weights, int num_tokens,
                   int topk, int hidden, int dtype, mi355q_stream_t stream);

/* ------------------------------------------------------------------ */
/* fp8 weight-only (W8A16) linear                                      */
/* ------------------------------------------------------------------ */
/*
 * C[M,N] = A[M,K] * (W * scale)^T (+ bias), fp32 accumulation, bias added in fp32, one
 * rounding (F.linear semantics).  W is the checkpoint tensor float8_e4m3fn [N, K] (fp8.py:319-322), untouched.
 * scales (fp32, device):
 *   block_k >= K (or <= 0): one scale per output channel, [N]            (per-tensor checkpoints
 *                           expand their per-logical-matrix scales to this at load time)
 *   otherwise             : [ceil(N/block_n), ceil(K/block_k)] block scales ("weight_scale_inv",
 *                           fp8.py:337-346), block_k a multiple of 16
 * The reference has no fp8 weight-only kernel on ROCm (kernels/linear/__init__.py:405-407);
 * the arithmetic is that of its dequantise-then-linear fallback (fp8.py:433-454) with the
 * weight kept in fp32 instead of being rounded to bf16.
 */
int mi355q_w8a16_gemm(const void* a, int lda, const void* w, const float* scales, const void* bias,
                      void* c, int M, int N, int K, int block_n, int block_k, int act_dtype,
                      mi355q_stream_t stream);
/* out[N,K] (act_dtype) = round(fp8(W) * scale): the weight of the dense-GEMM route for many
 * tokens (fp8.py:433-454). */
int mi355q_w8a16_dequantize(const void* w, const float* scales, void* out, int N, int K,
                            int block_n, int block_k, int act_dtype, mi355q_stream_t stream);

/*
 * 8-bit integer weight-only linear (GPTQ 8 bits, symmetric: scalar type uint8b128):
 * C[M,N] = A[M,K] * ((q - 128) * scale)^T (+ bias), same kernels and rounding as mi355q_w8a16_gemm.
 *   w      : uint8 [N, K], the checkpoint's values re-ordered from int32 [K/4, N] (four consecutive
 *            k per word, least significant byte first: auto_gptq.py:381-445 with pack_factor 4)
 *   scales : fp32 [N, K / group_k] (group_k = K: one scale per output channel), group_k a multiple of 16
 * Replaces ops.gptq_gemm with bit = 8 (vllm/_custom_ops.py:558-577; q_gemm.cu:1384-1419 with
 * qdq_8.cuh) and its reconstruct route (mi355q_wi8a16_dequantize + dense GEMM for many tokens).
 */
int mi355q_wi8a16_gemm(const void* a, int lda, const void* w, const float* scales, const void* bias,
                       void* c, int M, int N, int K, int group_k, int act_dtype, mi355q_stream_t stream);
int mi355q_wi8a16_dequantize(const void* w, const float* scales, void* out, int N, int K, int group_k,
                             int act_dtype, mi355q_stream_t stream);

/* ------------------------------------------------------------------ */
/* fp8 activations x fp8 weights (W8A8): quantisers and scaled product  */
/* ------------------------------------------------------------------ */
/*
 * All fp8 here is OCP float8_e4m3fn (gfx950), saturated to +-448 before round-to-nearest-even
 * (csrc/quantization/w8a8/fp8/common.cuh:57-77).  `dtype` is the input element type
 * (MI355Q_BF16 / _FP16; the three stand-alone quantisers also take MI355Q_FP32); strides count
 * elements; every scale is fp32 in device memory.
 *
 * out[t,n] = fp8(x[t,n] * (1 / scale[(t / group_m) * scale_stride_m + (n / group_n) * scale_stride_n])).
 * group_m / group_n <= 0 mean the whole extent; per tensor: strides 0, 0; per token: (1, -1) with
 * strides (1, 0); per channel: (-1, 1) with strides (0, 1); 2-D groups: the scale tensor's strides.
 * Replaces torch.ops._C.static_scaled_fp8_quant (vllm/_custom_ops.py:1832-1898;
 * csrc/libtorch_stable/quantization/w8a8/fp8/common.cu:11-76,183-335).
 */
int mi355q_static_scaled_fp8_quant(void* out, int64_t out_stride, const void* input, int64_t input_stride,
                                   const float* scale, int num_tokens, int hidden, int group_m, int group_n,
                                   int64_t scale_stride_m, int64_t scale_stride_n, int dtype,
                                   mi355q_stream_t stream);
/* scale[0] = max|x| / 448 over the whole tensor (written here: a memset node + a reduction kernel),
 * out = fp8(x * (1 / scale[0])).  Replaces torch.ops._C.dynamic_scaled_fp8_quant (common.cu:78-135,337-381). */
int mi355q_dynamic_scaled_fp8_quant(void* out, int64_t out_stride, const void* input, int64_t input_stride,
                                    float* scale, int num_tokens, int hidden, int dtype,
                                    mi355q_stream_t stream);
/* scales[t] = max(min(max|x[t,:]|, *scale_ub) / 448, 1 / (448 * 512)) (scale_ub may be NULL),
 * out[t,:] = fp8(x[t,:] / scales[t]).  Replaces torch.ops._C.dynamic_per_token_scaled_fp8_quant
 * (common.cu:137-179,383-422). */
int mi355q_dynamic_per_token_scaled_fp8_quant(void* out, int64_t out_stride, const void* input,
                                              int64_t input_stride, float* scales, const float* scale_ub,
                                              int num_tokens, int hidden, int dtype, mi355q_stream_t stream);
/* out[T,H] fp8 = fp8(float(round(x * rsqrt(mean(x^2) + eps) * w)) * (1 / scale[0])): RMSNorm whose result
 * passes through the activation dtype and is quantised with a given per-tensor scale.  Replaces
 * torch.ops._C.rms_norm_static_fp8_quant (csrc/libtorch_stable/layernorm_quant_kernels.cu:22-81,206-246). */
int mi355q_rms_norm_static_fp8_quant(void* out, const void* input, int64_t input_stride, const void* weight,
                                     const float* scale, float epsilon, int num_tokens, int hidden, int dtype,
                                     mi355q_stream_t stream);
/* residual[T,H] <- round(x + residual) first, then as above on the new residual.  Replaces
 * torch.ops._C.fused_add_rms_norm_static_fp8_quant (layernorm_quant_kernels.cu:87-202,264-309). */
int mi355q_fused_add_rms_norm_static_fp8_quant(void* out, const void* input, int64_t input_stride,
                                               void* residual, const void* weight, const float* scale,
                                               float epsilon, int num_tokens, int hidden, int dtype,
                                               mi355q_stream_t stream);
/* RMSNorm (after an optional residual add; residual may be NULL, else it receives round(x + residual))
 * quantised with one dynamic scale per token: v = float(x) + float(residual) unrounded,
 * y = round(round(v * rms) * w), scales[t] = max(min(max|y|, *scale_ub) / 448, 1 / (448 * 512)),
 * out = fp8(y / scales[t]).  Replaces torch.ops._C.rms_norm_dynamic_per_token_quant with an fp8 output
 * (vllm/_custom_ops.py:361-…; csrc/libtorch_stable/quantization/fused_kernels/
 * fused_layernorm_dynamic_per_token_quant.cu:11-91,174-206, layernorm_utils.cuh:17-228). */
int mi355q_rms_norm_dynamic_per_token_quant(void* out, float* scales, const void* input, int64_t input_stride,
                                            const void* weight, const float* scale_ub, void* residual,
                                            float epsilon, int num_tokens, int hidden, int dtype,
                                            mi355q_stream_t stream);
/* out[T,d] fp8 = fp8(float(round(round(silu(x[:, :d])) * x[:, d:])) * (1 / scale[0])), input [T, 2d]
 * contiguous.  Replaces torch.ops._C.silu_and_mul_quant
 * (csrc/libtorch_stable/quantization/activation_kernels.cu:34-109,584-…). */
int mi355q_silu_and_mul_quant(void* out, const void* input, const float* scale, int num_tokens, int d,
                              int dtype, mi355q_stream_t stream);
/*
 * C[M,N] (out_dtype) = round((A[M,K] . W[N,K]^T) * scale_a[m] * scale_b[n] + bias[n]): fp8 x fp8 products
 * accumulated in fp32 on the matrix cores.  A: fp8 [M, K], row stride lda; W: fp8 [N, K], row stride
 * ldw (the checkpoint's layout; the reference passes its transpose view); K, lda, ldw multiples of
 * 16.  scale_a: 1 element (per tensor) or M (per token); scale_b: 1 or N (per output channel).
 * bias: out_dtype [N] or NULL.  Up to 64 rows run on the weight-streaming kernel, more on the tiled one.
 * Replaces rocm_per_tensor_float_w8a8_scaled_mm_impl = ops.wvSplitKQ for <= 4 rows, else
 * torch._scaled_mm (vllm/model_executor/kernels/linear/scaled_mm/rocm.py:19-52; wvSplitKQ:
 * vllm/_custom_ops.py:2213-…, csrc/rocm/skinny_gemms.cu:1885-2345) and the per-token / per-channel
 * torch._scaled_mm routes (scaled_mm/pytorch.py).
 */
int mi355q_fp8_scaled_mm(const void* a, int64_t lda, const void* w, int64_t ldw, const float* scale_a,
                         int scale_a_numel, const float* scale_b, int scale_b_numel, const void* bias, void* c,
                         int64_t ldc, int M, int N, int K, int out_dtype, mi355q_stream_t stream);

/*
 * Per-token-group fp8 quantisation, the activations of the block-scaled product below: for every token and
 * every group of `group_size` consecutive elements, scale = max(eps, max |x|) / 448 and
 * out = fp8_e4m3fn(clamp(x / scale, -448, 448)).  input: bf16 / fp16 / fp32 [num_tokens, hidden], hidden a
 * multiple of group_size; the scale of (token t, group g) is written at scales[t * scale_stride_token +
 * g * scale_stride_group], which serves the row-major [T, H / gs] tensor and its column-major form alike.
 * Replaces per_token_group_quant_fp8 = torch.ops._C.per_token_group_fp8_quant
 * (vllm/model_executor/layers/quantization/utils/fp8_utils.py:533-659,
 * csrc/libtorch_stable/quantization/w8a8/fp8/per_token_group_quant.cu:40-160), without its ue8m0 scale rounding
 * (a DeepGEMM option that is off on ROCm).
 */
int mi355q_per_token_group_fp8_quant(void* out, int64_t out_stride, const void* input, int64_t input_stride,
                                     float* scales, int64_t scale_stride_token, int64_t scale_stride_group,
                                     int num_tokens, int hidden, int group_size, float eps, int dtype,
                                     mi355q_stream_t stream);
/*
 * Block-scaled fp8 product of DeepSeek-V3 / Qwen3 style checkpoints:
 *   C[m, n] (out_dtype) = round(sum over 128-wide k blocks b of (A[m, b] . W[n, b]) * a_scales[m, b] *
 *                         w_scales[n / block_n, b] + bias[n])
 * A: fp8 [M, K] (row stride lda) with one fp32 scale per (token, 128-wide block); W: fp8 [N, K] (row stride
 * ldw) with one fp32 scale per (block_n x block_k) tile; scale tensors are addressed through their strides.
 * block_k must be 128, block_n a multiple of 16 (of 32 beyond 64 rows); K, lda, ldw multiples of 16; a
 * ragged last block of K or N is allowed.  Every 128-wide block is accumulated on its own and folded into the
 * fp32 sum as (dot * a_scale) * w_scale, the order of the reference kernel.  Replaces
 * w8a8_triton_block_scaled_mm (fp8_utils.py:736-817,859-990) behind W8A8BlockFp8LinearOp.
 */
int mi355q_fp8_block_scaled_mm(const void* a, int64_t lda, const float* a_scales, int64_t a_scale_stride_token,
                               int64_t a_scale_stride_block, const void* w, int64_t ldw, const float* w_scales,
                               int64_t w_scale_stride_n, int64_t w_scale_stride_k, int block_n, int block_k,
                               const void* bias, void* c, int64_t ldc, int M, int N, int K, int out_dtype,
                               mi355q_stream_t stream);

/* ------------------------------------------------------------------ */
/* Unquantized skinny linear (lm_head, bf16 decode comparator)         */
/* ------------------------------------------------------------------ */
/* C[M,N] = A[M,K] * W[N,K]^T (+ bias); M <= 16 per pass.  Replaces
 * ops.wvSplitK / ops.LLMM1 (vllm/_custom_ops.py:2161-2213;
 * csrc/rocm/skinny_gemms.cu) as dispatched by layers/utils.py:116-186. */
int mi355q_linear_skinny(const void* a, int lda, const void* w, const void* bias, void* c, int M,
                         int N, int K, int act_dtype, mi355q_stream_t stream);

/* ------------------------------------------------------------------ */
/* Norms, RoPE, activation                                             */
/* ------------------------------------------------------------------ */
/* Replaces torch.ops._C.rms_norm (layernorm_kernels.cu:14-100,221-293). */
int mi355q_rms_norm(void* out, const void* input, int64_t input_stride, const void* weight,
                    float epsilon, int num_tokens, int hidden, int dtype, mi355q_stream_t stream);
/* In place on input and residual.  Replaces torch.ops._C.fused_add_rms_norm
 * (layernorm_kernels.cu:106-217,314-373). */
int mi355q_fused_add_rms_norm(void* input, int64_t input_stride, void* residual,
                              const void* weight, float epsilon, int num_tokens, int hidden,
                              int dtype, mi355q_stream_t stream);
/*
 * residual[t] = round(residual[t] + sum_s partial[s][t] (+ bias));
 * out[t] = round(residual[t] * rsqrt(mean(residual[t]^2) + eps) * weight).
 * The split-K partials of the preceding row-parallel GEMM are consumed here
 * instead of being reduced by a kernel of their own (no reference counterpart;
 * same arithmetic as gemm -> fused_add_rms_norm).
 */
int mi355q_splitk_add_rms_norm(void* out, void* residual, const float* partial, int splits,
                               const void* bias, const void* weight, float epsilon,
                               int num_tokens, int hidden, int dtype, mi355q_stream_t stream);
/* In place on query/key.  positions int64 [T].  Replaces ops.rotary_embedding
 * (vllm/_custom_ops.py:201-225; pos_encoding_kernels.cu:76-201). */
int mi355q_rotary_embedding(const int64_t* positions, void* query, int64_t query_stride,
                            void* key, int64_t key_stride, int head_size, int num_heads,
                            int num_kv_heads, const void* cos_sin_cache, int rot_dim,
                            int cache_dtype, int is_neox, int num_tokens, int dtype,
                            mi355q_stream_t stream);
/* out[T,d] = silu(x[T,:d]) * x[T,d:].  Replaces torch.ops._C.silu_and_mul. */
int mi355q_silu_and_mul(void* out, const void* input, int num_tokens, int d, int dtype,
                        mi355q_stream_t stream);
/* The same on a pair-interleaved row [T, 2 d] = (gate_0, up_0, gate_1, up_1, ...), the channel order a
 * gate_up weight repacked with MI355Q_W_PAIR_HALVES produces: out[t, i] = silu(in[t, 2 i]) * in[t, 2 i + 1].
 * d a multiple of 8. */
int mi355q_silu_and_mul_pairs(void* out, const void* input, int num_tokens, int d, int dtype,
                              mi355q_stream_t stream);

/* ------------------------------------------------------------------ */
/* KV cache + paged attention (ROCM_ATTN layout)                       */
/* K cache [blocks, kv_heads, head/x, block_size, x], x = 16 / elem    */
/* V cache [blocks, kv_heads, head, block_size]                        */
/* ------------------------------------------------------------------ */
/* kv_fp8 = 0: cache holds `dtype` elements; 1: OCP e4m3fn bytes of value/scale.
 * k_scale/v_scale: DEVICE fp32 scalars (read through the pointer, as the
 * reference does).  slot_mapping int64 [T], negative = skip.
 * Replaces ops.reshape_and_cache (vllm/_custom_ops.py:2592-2611;
 * cache_kernels.cu:254-312,749-780). */
int mi355q_reshape_and_cache(const void* key, const void* value, int64_t key_stride,
                             int64_t value_stride, void* key_cache, void* value_cache,
                             const int64_t* slot_mapping, int num_tokens, int num_kv_heads,
                             int head_size, int block_size, int64_t block_stride, int dtype,
                             int kv_fp8, const float* k_scale, const float* v_scale,
                             mi355q_stream_t stream);
/* The same write for the flash cache layouts: key_cache and value_cache both
 * [blocks, block_size, heads, head] (NHD) or [blocks, heads, block_size, head] (HND), described by the
 * element strides of the block, token-row (page) and head dimensions; k_scale / v_scale DEVICE fp32, one
 * value (scales_per_head = 0) or one per head (1).  slot_mapping int64 [T], negative = skip; T may be
 * smaller than the key tensor's first dimension (graph padding), as in the reference.
 * Replaces ops.reshape_and_cache_flash (vllm/_custom_ops.py:2614-2637;
 * cache_kernels.cu:314-400,795-867). */
int mi355q_reshape_and_cache_flash(const void* key, const void* value, int64_t key_stride,
                                   int64_t value_stride, void* key_cache, void* value_cache,
                                   const int64_t* slot_mapping, int num_tokens, int num_heads,
                                   int head_size, int block_size, int64_t block_stride,
                                   int64_t page_stride, int64_t head_stride, int dtype, int kv_fp8,
                                   const float* k_scale, const float* v_scale, int scales_per_head,
                                   mi355q_stream_t stream);
/* mi355q_rotary_embedding followed by mi355q_reshape_and_cache in one launch: q and k are rotated in
 * place, the rotated k and v are written to the paged cache (same layouts, same fp8 conversion; a
 * negative slot rotates the token without caching it).  Replaces the backend hook
 * do_rope_and_kv_cache_update (vllm/v1/attention/backends/rocm_attn.py:522-562, which calls AITER's
 * triton_rope_and_cache) and the op fused_rope_and_unified_kv_cache_update
 * (vllm/compilation/passes/fusion/rope_kvcache_fusion.py:46-96). */
int mi355q_rope_and_cache(const int64_t* positions, void* query, int64_t query_stride, void* key,
                          int64_t key_stride, const void* value, int64_t value_stride, int head_size,
                          int num_heads, int num_kv_heads, const void* cos_sin_cache, int rot_dim,
                          int cache_dtype, int is_neox, void* key_cache, void* value_cache,
                          const int64_t* slot_mapping, int num_tokens, int block_size,
                          int64_t block_stride, int dtype, int kv_fp8, const float* k_scale,
                          const float* v_scale, mi355q_stream_t stream);
/* The tail of a fused qkv projection computed in K-slices: partial fp32 [splits][T][N],
 * N = (heads + 2 kv_heads) * head_size.  Row t of qkv = round(sum of the slices) (+ bias, rounded
 * again), NeoX RoPE over the whole head on its q and k heads, k and v written to the paged cache: the
 * slice merge of mi355q_w4a16_mm followed by mi355q_rope_and_cache, in one launch.  No counterpart in
 * the reference (it runs gptq_gemm, rotary_embedding and reshape_and_cache one after the other). */
int mi355q_splitk_rope_and_cache(const float* partial, int splits, const void* bias, void* qkv,
                                 const int64_t* positions, int head_size, int num_heads, int num_kv_heads,
                                 const void* cos_sin_cache, int cache_dtype, void* key_cache,
                                 void* value_cache, const int64_t* slot_mapping, int num_tokens,
                                 int block_size, int64_t block_stride, int dtype, int kv_fp8,
                                 const float* k_scale, const float* v_scale, mi355q_stream_t stream);
/* dst = fp8 -> dtype or dtype -> fp8 element-wise with scale.  Replaces
 * ops.convert_fp8 (vllm/_custom_ops.py:2908; cache_kernels.cu:1003-1060). */
int mi355q_convert_fp8(void* dst, const void* src, int64_t n, float scale, int dtype,
                       int to_fp8, mi355q_stream_t stream);

#define MI355Q_PA_PARTITION 256
/*
 * Decode-only paged attention (rows whose query length != 1 are skipped when
 * query_start_loc != NULL).  Replaces ops.paged_attention_rocm
 * (vllm/_custom_ops.py:120-163; csrc/rocm/attention.cu:3652-3712): same
 * argument meaning; exp_sums/max_logits [S, H, P] fp32 and tmp_out [S, H, P, D]
 * are caller-allocated scratch with P = ceil(max_seq_len / 256).
 * The partitions of a long context are merged inside the same launch by the last
 * workgroup to arrive; its arrival counters are device memory owned by the library, zero
 * between launches, one bank per stream (the first 64 streams of a process that launch
 * attention; later streams, grids beyond a bank of 16384 counters, or MI355Q_PA_REDUCE_KERNEL=1
 * in the environment take the stateless variant with a separate merge launch).  A launch
 * captured into a HIP graph keeps the bank of its capture stream: graphs captured on one
 * stream must not be replayed concurrently.  This holds for the attention entry points below
 * as well.
 */
int mi355q_paged_attention(void* out, float* exp_sums, float* max_logits, void* tmp_out,
                           const void* query, int64_t q_stride, const void* key_cache,
                           const void* value_cache, int64_t kv_block_stride,
                           int64_t kv_head_stride, int num_seqs, int num_heads,
                           int num_kv_heads, int head_size, float scale,
                           const int32_t* block_tables, int max_blocks_per_seq,
                           const int32_t* seq_lens, const int32_t* query_start_loc,
                           int block_size, int max_seq_len, const float* alibi_slopes,
                           int dtype, int kv_fp8, const float* k_scale, const float* v_scale,
                           mi355q_stream_t stream);

/*
 * mi355q_paged_attention for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): the same kernels with exp_sums,
 * max_logits and tmp_out taken from a device array inside the library instead of from the caller.
 * No counterpart in the reference, whose op allocates these buffers through torch on every call
 * (chunked_prefill_paged_decode.py:379-395).  Calls nothing that allocates or frees.
 * Limit: num_seqs * num_heads * ceil(max_seq_len / 256) <= MI355Q_PA_NOALLOC_MAX_ITEMS
 * (MI355Q_ENOSPACE otherwise); head_size <= 128.  The array (about 4.1 MiB) exists once per GPU the
 * library runs on and is shared by every call on that GPU: calls that overlap on one GPU (different
 * streams) race on it.  Nothing in it has to be zero between calls.
 */
#define MI355Q_PA_NOALLOC_MAX_ITEMS 16384
int mi355q_paged_attention_noalloc(void* out, const void* query, int64_t q_stride, const void* key_cache,
                                   const void* value_cache, int64_t kv_block_stride,
                                   int64_t kv_head_stride, int num_seqs, int num_heads, int num_kv_heads,
                                   int head_size, float scale, const int32_t* block_tables,
                                   int max_blocks_per_seq, const int32_t* seq_lens,
                                   const int32_t* query_start_loc, int block_size, int max_seq_len,
                                   const float* alibi_slopes, int dtype, int kv_fp8, const float* k_scale,
                                   const float* v_scale, mi355q_stream_t stream);

/*
 * Attention for rows with query length >= 1 (prompt rows), reading K/V back from
 * the paged cache after reshape_and_cache has written the new tokens: query row
 * j of sequence s attends cache positions [0, seq_lens[s] - query_len[s] + j].
 * query [T, H, D]; query_start_loc int32 [S+1] (required); scratch buffers are
 * indexed by query row: exp_sums/max_logits [T, H, P], tmp_out [T, H, P, D].
 * Stands in for the prompt rows that the reference hands to context_attention_fwd
 * (vllm/v1/attention/ops/chunked_prefill_paged_decode.py:303-326,
 * ops/prefix_prefill.py).
 */
int mi355q_paged_prefill_attention(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* query,
    int64_t q_stride, const void* key_cache, const void* value_cache, int64_t kv_block_stride,
    int64_t kv_head_stride, int num_seqs, int num_heads, int num_kv_heads, int head_size,
    float scale, const int32_t* block_tables, int max_blocks_per_seq, const int32_t* seq_lens,
    const int32_t* query_start_loc, int max_query_len, int block_size, int max_seq_len,
    const float* alibi_slopes, int dtype, int kv_fp8, const float* k_scale, const float* v_scale,
    mi355q_stream_t stream);

/*
 * One decode step of attention in a single launch: NeoX RoPE (rotary_dim ==
 * head_size) on the query and on the new key, write of the new key/value into
 * the paged cache at position seq_lens[s] - 1, and paged attention over
 * [0, seq_lens[s]).  qkv [S, (H + 2 KVH) * D] is the fused qkv projection
 * output (q | k | v per row, left untouched); cos_sin_cache [max_pos, D] in
 * `dtype`.  Same arithmetic as mi355q_rotary_embedding +
 * mi355q_reshape_and_cache + mi355q_paged_attention (no reference counterpart
 * as one op; the reference runs ops.rotary_embedding, ops.reshape_and_cache and
 * ops.paged_attention_rocm back to back, rocm_attn.py:367-517).
 */
int mi355q_decode_attention_fused(
    void* out, float* exp_sums, float* max_logits, void* tmp_out, const void* qkv,
    int64_t qkv_stride, const int64_t* positions, const void* cos_sin_cache, void* key_cache,
    void* value_cache, int64_t kv_block_stride, int64_t kv_head_stride, int num_seqs,
    int num_heads, int num_kv_heads, int head_size, float scale, const int32_t* block_tables,
    int max_blocks_per_seq, const int32_t* seq_lens, int block_size, int max_seq_len, int dtype,
    int kv_fp8, const float* k_scale, const float* v_scale, mi355q_stream_t stream);

/* ------------------------------------------------------------------ */
/* Sampler inner loop                                                  */
/* ------------------------------------------------------------------ */
/* ids[b] = first index of max(float(logits[b, :])).  Replaces
 * logits.float().argmax(-1) (vllm/v1/sample/sampler.py:96,240-241).
 * Rows of 16384 values or more are cut into segments whose results meet in one 64-bit
 * slot per row (library-owned device memory, zero between launches, up to 1024 rows; more
 * rows, unaligned rows or MI355Q_ARGMAX_NARROW=1 take the single-workgroup kernel).  Argmax
 * launches that overlap on one GPU would share the slots: keep them on one stream. */
int mi355q_argmax(const void* logits, int64_t row_stride, int64_t* ids, int num_rows, int vocab,
                  int dtype, mi355q_stream_t stream);

/* mi355q_argmax plus what a greedy decode step does with the sampled id, in the same launch: positions[b] += 1,
 * seq_lens[b] += 1 (either may be NULL) and hidden[b, :] = embedding[ids[b], :] (16-bit rows; both NULL: no
 * gather) -- the next step's input, so that a captured step needs no separate increments or embedding lookup.
 * The id must be a valid row of `embedding`.  No single reference counterpart: the reference runs
 * logits.argmax (sampler.py:240-241), the model runner's position / sequence-length bookkeeping
 * (gpu_model_runner.py:2001) and VocabParallelEmbedding as separate steps. */
int mi355q_argmax_decode_step(const void* logits, int64_t row_stride, int64_t* ids, int num_rows, int vocab,
                              int dtype, int64_t* positions, int32_t* seq_lens, const void* embedding,
                              int64_t embedding_row_stride, void* hidden, int64_t hidden_row_stride,
                              int hidden_size, mi355q_stream_t stream);

/* Random sampling with top-k / top-p: ids[b] = argmax_i softmax(mask(logits[b] / temperature[b]))[i] / noise[b, i],
 * first index on ties.  mask keeps the top_k[b] largest logits (ties with the k-th kept; top_k NULL,
 * <= 0 or >= vocab: no top-k), then drops the longest ascending prefix whose cumulative probability
 * stays <= 1 - top_p[b] and never the largest token (top_p NULL or >= 1: no top-p).  noise holds the
 * caller's Exp(1) draws (torch's exponential_ under the request generators), so the random stream stays
 * the caller's.  fp32 logits, one workgroup per row, thresholds by radix selection (no sort).
 * Replaces apply_top_k_top_p_pytorch + random_sample (vllm/v1/sample/ops/topk_topp_sampler.py:367-407,
 * 441-466) behind Sampler.sample (vllm/v1/sample/sampler.py:228-302).  Tokens with equal logits at a
 * threshold are kept or dropped together (the reference orders them by an unspecified sort). */
int mi355q_sample_topk_topp(const float* logits, int64_t row_stride, const float* temperature,
                            const int32_t* top_k, const float* top_p, const float* noise, int64_t* ids,
                            int num_rows, int vocab, mi355q_stream_t stream);
/* In place; masks are bool (1 byte) [B, V]; penalties [B] in logits dtype.
 * Replaces ops.apply_repetition_penalties_ (vllm/_custom_ops.py:325-333;
 * sampler.cu:13-49,618-659). */
int mi355q_apply_repetition_penalties(void* logits, const uint8_t* prompt_mask,
                                      const uint8_t* output_mask, const void* penalties,
                                      int num_seqs, int vocab, int dtype, mi355q_stream_t stream);

/* ------------------------------------------------------------------ */
/* One-shot all-reduce over peer memory (tensor-parallel decode)       */
/* ------------------------------------------------------------------ */
/*
 * Sum of a small 16-bit tensor over the ranks of a tensor-parallel group, one process per GPU, through
 * regions of uncached device memory that every rank maps from every peer (xGMI): a call pushes the
 * input into all regions as 8-byte {data, epoch} granules, sweeps its own region until every rank's
 * granules carry the epoch, and adds them in rank order in fp32 (all ranks compute the same bits).  One
 * kernel, no host synchronisation, capturable; the epoch is device state advanced by the kernel.
 * Replaces the reference's custom / quick all-reduce for the [tokens, hidden] sums behind o_proj and
 * down_proj (csrc/libtorch_stable/custom_all_reduce.cu:165-200, csrc/custom_all_reduce.cuh:291-305,
 * csrc/quickreduce/quick_reduce.h:117-215; Python side vllm/distributed/device_communicators/
 * quick_all_reduce.py:44-62, cuda_communicator.py:275-340).  Life cycle as the reference's:
 *   init (allocates the region; 2..8 ranks; max_bytes a multiple of 16 up to 8 MiB: the largest tensor)
 *   -> get_handle (mi355q_peer_ar_handle_bytes() bytes) -> exchange the handles by any means
 *   -> open_handles (all ranks' handles in rank order; this rank's own entry is ignored)
 *   -> all_reduce ... -> destroy.
 * all_reduce: num_elements even, at most max_bytes; input may equal output.  Calls on one communicator must
 * be stream-ordered, and every rank must make the same sequence of calls.  A rank that waits longer than
 * MI355Q_PEER_AR_TIMEOUT_MS (default 5000) for a peer counts a timeout and returns garbage instead of
 * hanging: mi355q_peer_ar_timeouts synchronises the device and reports the count.
 */
int mi355q_peer_ar_handle_bytes(void);
int mi355q_peer_ar_init(int rank, int world_size, size_t max_bytes, void** comm_out);
int mi355q_peer_ar_get_handle(void* comm, void* handle_out);
int mi355q_peer_ar_open_handles(void* comm, const void* handles);
size_t mi355q_peer_ar_max_bytes(void* comm);
int mi355q_peer_ar_all_reduce(void* comm, const void* input, void* output, size_t num_elements, int dtype,
                              mi355q_stream_t stream);
/* The same exchange without the sum: output [rows, world * row_elements] holds, in every row, the ranks' rows side by
 * side in rank order (the vocabulary-parallel logits before sampling; replaces the all-gather of
 * cuda_communicator.py:342-390 at decode sizes).  rows * row_elements * 2 <= max_bytes; not in place. */
int mi355q_peer_ar_all_gather(void* comm, const void* input, void* output, int rows, size_t row_elements, int dtype,
                              mi355q_stream_t stream);
int mi355q_peer_ar_timeouts(void* comm, int* timeouts_out);
int mi355q_peer_ar_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* MI355Q_H */
