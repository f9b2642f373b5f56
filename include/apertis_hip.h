/*
 * apertis_hip.h — C ABI of libapertis_hip.so, the MI355X (gfx950) kernel library behind the
 * Apertis hot path (selective-SSM scan, MoE router/permute/grouped-GEMM, patch-embed GEMM).
 *
 * The reference (CuzImSlymi/Apertis-LLM) is pure Python and has no FFI of its own; each entry
 * point below replaces a chain of torch ops inside one reference function, cited per function
 * as /root/reference/<file>:<lines>.  INTEGRATION.md shows the ctypes stubs a reference
 * maintainer would add.
 *
 * Conventions (every function):
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted;
 *   - the caller owns every buffer (inputs, outputs, workspaces); kernels never allocate;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it, no sync;
 *   - returns 0 (APERTIS_OK) or a negative error code; never throws across the ABI;
 *   - re-entrant, safe to dlopen lazily after fork().  No mutable process-global state: kernels keep nothing between
 *     launches (cross-work-group counters live in caller-owned workspaces); the only cached value is a device attribute
 *     (the CU count, read once).
 */
#ifndef APERTIS_HIP_H
#define APERTIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APERTIS_OK 0
#define APERTIS_ERR_ARG (-1)         /* null pointer / negative size / bad enum            */
#define APERTIS_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for       */
#define APERTIS_ERR_LAUNCH (-3)      /* hipGetLastError() after a launch was not success   */

#define APERTIS_F32 0
#define APERTIS_BF16 1

#define APERTIS_ACT_NONE 0
#define APERTIS_ACT_GELU 1 /* erf GELU, torch.nn.GELU() default */
#define APERTIS_ACT_RELU 2
#define APERTIS_ACT_SILU 3
/* Flags of apertis_grouped_gemm_nt's `act` (bf16, GELU, shapes the two-per-CU kernel takes: K <= 1024, N >= 512, rows >= 4096;
 * anything else returns APERTIS_ERR_UNSUPPORTED and the caller keeps the pre-activation form):
 *   forward  act | APERTIS_ACT_SAVE_GRAD: `pre_act` receives g' = act'(pre) * keep / (1-p) instead of pre;
 *   dgrad    act = APERTIS_ACT_MUL_SAVED with `act_bwd_pre` = that g': out = (A W^T) * g' - one multiply per element in
 *            the epilogue instead of the activation derivative and the mask hash. */
#define APERTIS_ACT_SAVE_GRAD 0x100
#define APERTIS_ACT_MUL_SAVED 0x200
/* with APERTIS_ACT_SAVE_GRAD: ask for the interleaved-epilogue kernel (grouped_gemm_nt2i_k, round 6: one wave per SIMD, the
 * epilogue of tile i between the MFMA groups of tile i + 1; K % 32 == 0, K >= 512, no tile queue).  Same bits as the default
 * kernel; measured 15 % SLOWER at the bench shape (profiles/r6_probe_nt2i_vs_nt4r.log), so no caller sets it by default. */
#define APERTIS_ACT_INTERLEAVED 0x400

/* Library/ABI version: (major<<16)|minor.  Bumped when a signature changes or an entry point is added (round 6: 4.7 - apertis_cross_entropy_fwd_bwd, apertis_layernorm_combine_bwd; round 5: 4.5 - apertis_scan_lookback_*, apertis_tiny_linear_bwd_pad; round 4: 4.4 - lean scan
 * entry points, apertis_scan_lean_fwd_dt, apertis_grouped_gemm_tn_dense_variant, apertis_weight_prep, apertis_ssm_decode_state_dt).  A host binding should
 * refuse a library whose version differs from the header it was written against (apertis_llm_amd/_lib.py does). */
#define APERTIS_ABI_VERSION ((4 << 16) | 7)
int apertis_abi_version(void);
/* Name of the code-object architecture this library was compiled for ("gfx950"). */
const char *apertis_arch(void);
/* Human-readable text for an error code returned by any function below. */
const char *apertis_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Selective-SSM scan  (replaces SelectiveLinearAttention._ssm_pytorch_scan_recurrent,
 * src/model/core.py:337-353, the semantic ground truth; and _ssm_scan_parallel :324-335)
 *
 *   A[c]    = -exp(A_log[c])                      c = head*N + n   (h*N = Dn channels)
 *   delta   = softplus(dlt) if delta_softplus else dlt      (core.py:383)
 *   a[b,t,c]= exp(delta[b,t,head(c)] * A[c])
 *   s[b,t,c]= a[b,t,c]*s[b,t-1,c] + Bt[b,t,c]     s[b,-1,c] = h0[b,c] or 0
 *   y[b,t,c]= C[b,t,c]*s[b,t,c]
 *
 * Layout: token-major.  dlt is [B,L,h] fp32 (row stride h).  Bt, C are [B,L,Dn] views with an
 * explicit row stride in ELEMENTS (they are column slices of the x_param_proj output,
 * core.py:377-385) and batch stride L*row_stride.  y is [B,L,Dn] with row stride y_rs.
 * dtype_bc / dtype_y select fp32 or bf16 storage; state and arithmetic are always fp32.
 *
 * Chunked scan: the workspaces are laid out in chunks of apertis_scan_chunk_len() tokens (the backward's chunk length;
 * the forward walks pairs of them and saves the carry-in of each).
 *   agg   : workspace  [B, nchunks, Dn, 2] fp32   (chunk aggregates, scratch)
 *   h_in  : output     [B, nchunks, Dn]    fp32   (state entering each chunk; saved for bwd)
 *   h_last: optional   [B, Dn] fp32 final state (core.py:351 new_ssm_state), may be NULL
 * ------------------------------------------------------------------------------------------ */
int64_t apertis_scan_chunk_len(int64_t B, int64_t L, int64_t Dn);
int64_t apertis_scan_num_chunks(int64_t B, int64_t L, int64_t Dn);

int apertis_selective_scan_fwd(const float *dlt, const float *A_log,
                               const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                               const float *h0, void *y, int64_t y_rs, float *h_last,
                               float *agg, float *h_in,
                               int64_t B, int64_t L, int64_t h, int64_t N,
                               int dtype_bc, int dtype_y, int delta_softplus, void *stream);

/* Backward of the scan (no reference code: autograd through core.py:347-349).
 *   dy            : [B,L,Dn] (row stride dy_rs, dtype_y)
 *   dBt, dC       : [B,L,Dn] views (row strides dbt_rs/dc_rs, dtype_bc)
 *   d_dlt         : [B,L,h] fp32; gradient w.r.t. dlt (through softplus if delta_softplus)
 *   dA_log        : [h*N] fp32 (overwritten)
 *   h_in          : saved by the forward
 *   agg           : workspace [B,nchunks,Dn,2] fp32
 *   mu_in         : workspace [B,nchunks,Dn] fp32, required (its head holds the row-group sums of the two-level dA_log fold)
 *   dA_part       : workspace [B*nchunks, Dn]  fp32
 */
int apertis_selective_scan_bwd(const float *dlt, const float *A_log,
                               const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                               const void *dy, int64_t dy_rs, const float *h_in,
                               void *dBt, int64_t dbt_rs, void *dC, int64_t dc_rs,
                               float *d_dlt, float *dA_log,
                               float *agg, float *mu_in, float *dA_part,
                               int64_t B, int64_t L, int64_t h, int64_t N,
                               int dtype_bc, int dtype_y, int delta_softplus, void *stream);

/* Post-scan gate (core.py:395-396):  out = (y + D[c]*xc) * silu(z),  all [T,Dn] row-major
 * with row strides; dtype_io for xc/z/out, dtype_y for y.  Backward returns dy, dxc, dz, dD. */
int apertis_ssm_gate_fwd(const void *y, int64_t y_rs, const void *xc, int64_t xc_rs,
                         const void *z, int64_t z_rs, const float *D, void *out, int64_t out_rs,
                         int64_t T, int64_t Dn, int dtype_y, int dtype_io, void *stream);
int apertis_ssm_gate_bwd(const void *dout, int64_t dout_rs, const void *y, int64_t y_rs,
                         const void *xc, int64_t xc_rs, const void *z, int64_t z_rs,
                         const float *D, void *dy, int64_t dy_rs, void *dxc, int64_t dxc_rs,
                         void *dz, int64_t dz_rs,
                         float *dD_part /* workspace [apertis_ssm_gate_bwd_blocks, Dn] */,
                         float *dD /* [Dn] out */, int64_t T, int64_t Dn, int dtype_y,
                         int dtype_io, void *stream);
int64_t apertis_ssm_gate_bwd_blocks(int64_t T, int64_t Dn);

/* ------------------------------------------------------------------------------------------
 * Scan with the skip + gate fused in  (SelectiveLinearAttention.forward, src/model/core.py:388-396:
 * the recurrence :337-353 followed by  out = (y + D*xc) * silu(z)  :395-396).  y never reaches HBM; the
 * backward recomputes it from the states it rebuilds.  One activation dtype for Bt, C, xc, z, out and their gradients.
 *
 * Algorithmic bytes per token ("fused epilogue variant"; e = bytes of the activation dtype):
 *   forward 5*Dn*e + 4h,  backward 9*Dn*e + 8h.
 *
 *   single_pass = 0: two launches (state pass + replay), `agg` required;
 *   single_pass = 1: ONE launch - work-groups take their chunk from a ticket counter, publish the chunk aggregate as
 *     8-byte {epoch, value} granules and gather the earlier chunks' aggregates while the rest of their tiles load;
 *     same bits as single_pass = 0.  Needs `ws` (apertis_scan_gate_workspace_bytes() bytes, zero-filled ONCE by the
 *     caller, then reused by ONE stream at a time) and `epoch` (non-zero, incremented by exactly 1 per launch that uses
 *     `ws`: the two ticket counters in its head alternate).  The int32 at byte 8 of `ws` is an error word (non-zero: a
 *     bounded wait timed out and the outputs of that launch are invalid).
 *   agg, h_in, h_last, h0: as for apertis_selective_scan_fwd (h_in is saved per apertis_scan_gate_chunk_len() tokens;
 *   nchunks here = ceil(L / apertis_scan_gate_chunk_len())).
 * Backward:
 *   dout [B,L,Dn]; dBt, dC [B,L,store_w] with Dn <= store_w <= ceil(Dn/64)*64: columns [Dn, store_w) are written as
 *   zeros (the zero-padded slices of the projection output's gradient buffer); dxc, dz [B,L,Dn]; d_dlt [B,L,h] fp32;
 *   dA_dD [2*Dn] fp32 = (dA_log | dD), overwritten; part: workspace [B*nchunks, 2*Dn] fp32; fold: workspace
 *   [64, 2*Dn] fp32.
 * ------------------------------------------------------------------------------------------ */
int64_t apertis_scan_gate_workspace_bytes(int64_t B, int64_t L, int64_t Dn);
/* Tokens per work item of the fused kernels: h_in, agg and the backward's partial sums (`part`: [B * chunks][2 * Dn]) are
 * laid out per chunk of this many tokens (ceil(L / len) chunks per sequence). */
int64_t apertis_scan_gate_chunk_len(void);
int apertis_scan_gate_fwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                          const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const float *h0,
                          void *out, int64_t out_rs, float *h_last, float *agg, float *h_in, void *ws, uint32_t epoch,
                          int64_t B, int64_t L, int64_t h, int64_t N, int dtype, int delta_softplus, int single_pass,
                          void *stream);
int apertis_scan_gate_bwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                          const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const void *dout,
                          int64_t dout_rs, const float *h_in, void *dBt, int64_t dbt_rs, void *dC, int64_t dc_rs,
                          int64_t store_w, void *dxc, int64_t dxc_rs, void *dz, int64_t dz_rs, float *d_dlt, float *dA_dD,
                          float *agg, float *fold, float *part, void *ws, uint32_t epoch, int64_t B, int64_t L, int64_t h,
                          int64_t N, int dtype, int delta_softplus, int single_pass, void *stream);

/* Lean forms of the fused scan + gate (round 4; reference core.py:337-353,394-397; csrc/scan_gate.hip "LEAN"): a lane owns four
 * channels of a row, a wave one 64-token item, nothing is staged through LDS; state pass + chunk prefix + replay as three
 * launches per direction.  Shapes taken: bf16, N = 16, 128 < h*N <= 256, row strides and pointers multiples of 8 bytes, every
 * tensor below 4 GiB - APERTIS_ERR_UNSUPPORTED otherwise (the caller then uses apertis_scan_gate_fwd / _bwd, whose buffers
 * `h_in` / `h_last` mean the same).  Scratch is the caller's: agg [B, nchunks, h*N, 2] fp32 and, backward, mu_in [B, nchunks,
 * h*N] fp32, fold / part as for apertis_scan_gate_bwd.  `ckpt` [B, ceil(L/4), h*N] fp32 (forward: may be NULL) receives the
 * state entering every fourth token; the lean backward rebuilds its states from it and REQUIRES the one its forward wrote. */
int apertis_scan_lean_fwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                          const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const float *h0, void *out,
                          int64_t out_rs, float *h_last, float *agg, float *h_in, float *ckpt, int64_t B, int64_t L, int64_t h,
                          int64_t N, int delta_softplus, void *stream);
/* The same forward with dt_proj_head inside its state pass (N4's pre-scan prologue; core.py:382-383 - replaces a
 * apertis_tiny_linear_fwd launch): the delta logits are formed from the dt columns of the projection output (`dt_in`:
 * [B*L rows, R bf16] at row stride dt_rs elements, 16-byte aligned, dt_rs % 8 == 0, whole 16-byte chunks readable - the
 * columns behind R up to the next multiple of 8 must hold finite values) with W_dt [h, R] fp32 and b_dt [h] fp32 (may be
 * NULL), in apertis_tiny_linear_fwd's accumulation order (the same bits), and are WRITTEN to dlt [B, L, h] for the backward
 * (apertis_scan_lean_bwd / apertis_tiny_linear_bwd take them from there).  R <= 64, h <= 16. */
int apertis_scan_lean_fwd_dt(const void *dt_in, int64_t dt_rs, const float *W_dt, const float *b_dt, int64_t R, float *dlt,
                             const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs, const void *xc,
                             int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const float *h0, void *out,
                             int64_t out_rs, float *h_last, float *agg, float *h_in, float *ckpt, int64_t B, int64_t L,
                             int64_t h, int64_t N, int delta_softplus, void *stream);
int apertis_scan_lean_bwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                          const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const void *dout,
                          int64_t dout_rs, const float *ckpt, void *dBt, int64_t dbt_rs, void *dC, int64_t dc_rs, int64_t store_w,
                          void *dxc, int64_t dxc_rs, void *dz, int64_t dz_rs, float *d_dlt, float *dA_dD, float *agg, float *mu_in,
                          float *fold, float *part, int64_t B, int64_t L, int64_t h, int64_t N, int delta_softplus, void *stream);
/* The fused scan + gate as ONE launch per direction with a decoupled look-back (round 5; reference core.py:337-353,394-397;
 * csrc/scan_lookback.hip): the lean layout (a lane owns four channels of a row), a work-group per 64-token chunk with 16 tokens
 * per wave held in registers, so every operand row is read once; chunk aggregates and, per four chunks, an inclusive state are
 * published as {epoch, value} granules in `ws` and gathered by later chunks of the same sequence (work-groups take their chunk
 * from a ticket counter in chunk-major order, so a wait only ever concerns a work-group that has started).  Shapes taken: bf16,
 * N = 16, h*N <= 256 (several sequences side by side in a wave when h*N <= 128), row strides and pointers multiples of 8 bytes,
 * every tensor below 4 GiB - APERTIS_ERR_UNSUPPORTED otherwise (the caller then uses the forms above).
 *   ws     apertis_scan_lookback_workspace_bytes() bytes, 16-byte aligned, zero-filled ONCE by the caller and then only ever
 *          touched by these two functions and apertis_scan_gate_fwd / _bwd (the head - ticket counters, error word - is shared);
 *          `epoch` as for those: starts at 1 and grows by exactly one per launch on that workspace.  A look-back wait that
 *          times out ORs 2 into the error word (int at byte 8 of ws); the launch's outputs are then invalid.
 *   h_in   [B, nchunks, h*N] fp32 or NULL: the state entering every 64-token chunk (what apertis_scan_gate_bwd consumes)
 *   ckpt16 [B, ceil(L/16), h*N] fp32 (forward: may be NULL): the state entering every 16th token; apertis_scan_lookback_bwd
 *          rebuilds its states from it and REQUIRES the one its forward wrote.  (ckpt16[:, 4*j] is h_in[:, j].)
 *   fold / part, dA_dD, store_w and the gradient slices as for apertis_scan_gate_bwd. */
int64_t apertis_scan_lookback_workspace_bytes(int64_t B, int64_t L, int64_t Dn);
int apertis_scan_lookback_fwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                              const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const float *h0,
                              void *out, int64_t out_rs, float *h_last, float *h_in, float *ckpt16, void *ws, uint32_t epoch,
                              int64_t B, int64_t L, int64_t h, int64_t N, int delta_softplus, void *stream);
int apertis_scan_lookback_bwd(const float *dlt, const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs,
                              const void *xc, int64_t xc_rs, const void *z, int64_t z_rs, const float *D, const void *dout,
                              int64_t dout_rs, const float *ckpt16, void *dBt, int64_t dbt_rs, void *dC, int64_t dc_rs,
                              int64_t store_w, void *dxc, int64_t dxc_rs, void *dz, int64_t dz_rs, float *d_dlt, float *dA_dD,
                              float *fold, float *part, void *ws, uint32_t epoch, int64_t B, int64_t L, int64_t h, int64_t N,
                              int delta_softplus, void *stream);
/* Single-token decode step, re-ordered (csrc/decode_step.hip; core.py:364-400 with L = 1, called by generate() :1578-1603): the
 * reference keeps the FIRST conv output of [cached window | new xp] - which sees window[0] only -, so the conv output, x_param_proj,
 * the dt projection, the state update and  pre = C s + D xc  of EVERY layer depend on the caches alone and run at the start of
 * the token step, the NL layers as one more batch dimension:
 *   apertis_decode_pre_conv   conv_state [NL,B,Dn,k-1], w [NL,Dn,k], bias [NL,Dn] -> xc [NL,B,Dn]        (apertis_ssm_decode_conv's value)
 *   (x_param_proj of all layers: apertis_grouped_gemm_nt with one group per layer)
 *   apertis_decode_pre_state  p [NL*B, p_rs] (columns off_bt / off_c: Bt, C of width h*N; off_dt: the R dt columns), W_dt [NL,h,R],
 *                             b_dt [NL,h] or NULL, A_log [NL,h,N], D [NL,h*N], xc -> state [NL,B,h*N] fp32 in place, pre [NL,B,h*N] fp32
 * and per layer, between in_proj and out_proj:
 *   apertis_decode_post       pre [B,Dn], xz [B, >= 2 Dn] (xp | z) -> gated [B,Dn] = pre * silu(z); conv_state [B,Dn,k-1] shifted by xp
 * Same arithmetic as apertis_ssm_decode_conv + apertis_ssm_decode_state_dt: bit-identical values and caches.  2 <= k <= 16. */
int apertis_decode_pre_conv(const void *conv_state, const float *w, const float *bias, void *xc, int64_t NL,
                            int64_t B, int64_t Dn, int64_t k, int dtype, void *stream);
int apertis_decode_pre_state(const void *p, int64_t p_rs, int64_t off_bt, int64_t off_c, int64_t off_dt,
                             const float *W_dt, const float *b_dt, int64_t R, const float *A_log,
                             const float *D, const void *xc, float *state, float *pre, int64_t NL, int64_t B,
                             int64_t h, int64_t N, int delta_softplus, int dtype, void *stream);
int apertis_decode_post(const float *pre, const void *xz, int64_t xz_rs, void *conv_state, void *gated,
                        int64_t B, int64_t Dn, int64_t k, int dtype, void *stream);
/* Single-token decode step of the SSM block (core.py:364-400 with L = 1 and a cache, called from generate()
 * core.py:1578-1603), two kernels around the caller's x_param_proj / dt projections:
 *   apertis_ssm_decode_conv : window = [conv_state (k-1 tokens) | xp]; xc = silu(w[:, k-1]*window[0] + bias) - the
 *     reference keeps the FIRST output of the padded conv over that window (core.py:369-373), reproduced as is;
 *     conv_state_out = the last k-1 tokens of the window.  xp [B,Dn] (row stride xp_rs), conv_state / conv_state_out
 *     [B,Dn,k-1] (may be the same buffer), w [Dn,k] fp32, bias [Dn] fp32, xc [B,Dn] contiguous.
 *   apertis_ssm_decode_state: s = exp(delta*A)*s + Bt (state [B,Dn] fp32, updated IN PLACE), out = (C*s + D*xc)*silu(z);
 *     dt_logits [B,h] fp32, Bt / C / z [B,Dn] with row strides, xc / out [B,Dn] contiguous. */
int apertis_ssm_decode_conv(const void *xp, int64_t xp_rs, const void *conv_state, void *conv_state_out, const float *w,
                            const float *bias, void *xc, int64_t B, int64_t Dn, int64_t k, int dtype, void *stream);
int apertis_ssm_decode_state(const float *dt_logits, const float *A_log, const void *Bt, int64_t bt_rs, const void *C,
                             int64_t c_rs, const void *xc, const void *z, int64_t z_rs, const float *D, float *state,
                             void *out, int64_t B, int64_t h, int64_t N, int dtype, int delta_softplus, void *stream);
/* The same with dt_proj_head inside (core.py:382: one launch less per layer of a token step): the delta logits are formed from
 * dt_in [B, R] (the dt columns of the x_param_proj output, dtype as the other operands, row stride dt_rs elements), W_dt [h, R]
 * and b_dt [h] (may be NULL) fp32, in apertis_tiny_linear_fwd's accumulation order - the same bits. */
int apertis_ssm_decode_state_dt(const void *dt_in, int64_t dt_rs, const float *W_dt, const float *b_dt, int64_t R,
                                const float *A_log, const void *Bt, int64_t bt_rs, const void *C, int64_t c_rs, const void *xc,
                                const void *z, int64_t z_rs, const float *D, float *state, void *out, int64_t B, int64_t h,
                                int64_t N, int dtype, int delta_softplus, void *stream);

/* Residual + dropout of every sub-block (core.py:836-837, 918-919): y = res + keep/(1-p) * x over
 * n elements (n % 4 == 0); mask = counter hash of (seed, element index).  Backward: dx = keep/(1-p)*g
 * (the residual's gradient is g itself). */
int apertis_dropout_add_fwd(const void *x, const void *res, void *y, int64_t n, float drop_p,
                            uint64_t seed, int dtype_x, int dtype_res, void *stream);
int apertis_dropout_bwd(const void *g, void *dx, int64_t n, float drop_p, uint64_t seed,
                        int dtype_g, int dtype_x, void *stream);

/* Depthwise causal conv1d (k taps, left pad k-1, keep first L) + SiLU on token-major data
 * (replaces core.py:368-375: transpose -> nn.Conv1d(groups=Dn, padding=k-1)[:, :, :L] ->
 * transpose -> F.silu).  x,out: [B,L,Dn] with row strides; w: [Dn,k] fp32; bias: [Dn] fp32. */
int apertis_dwconv_silu_fwd(const void *x, int64_t x_rs, const float *w, const float *bias,
                            void *out, int64_t out_rs, int64_t B, int64_t L, int64_t Dn,
                            int64_t k, int dtype_io, void *stream);
int apertis_dwconv_silu_bwd(const void *x, int64_t x_rs, const float *w, const float *bias,
                            const void *dout, int64_t dout_rs, void *dx, int64_t dx_rs,
                            float *dw_part /* workspace [nblk,Dn,k] */,
                            float *db_part /* workspace [nblk,Dn] */,
                            float *dw /* [Dn,k] out */, float *db /* [Dn] out */,
                            int64_t B, int64_t L, int64_t Dn, int64_t k, int dtype_io,
                            void *stream); /* nblk = apertis_dwconv_bwd_blocks(B,L,Dn) */
int64_t apertis_dwconv_bwd_blocks(int64_t B, int64_t L, int64_t Dn);
/* apertis_dwconv_silu_bwd with a SECOND gradient of the same output (dout2, row stride dout2_rs; NULL: none): the conv's
 * output feeds x_param_proj and the scan (core.py:376,388-396), and the sum of their two gradients - rounded to the io dtype as
 * the framework's add kernel rounds it - is formed where the rows are read instead of in a [B, L, Dn] pass of its own. */
int apertis_dwconv_silu_bwd2(const void *x, int64_t x_rs, const float *w, const float *bias,
                             const void *dout, int64_t dout_rs, const void *dout2, int64_t dout2_rs,
                             void *dx, int64_t dx_rs, float *dw_part, float *db_part, float *dw,
                             float *db, int64_t B, int64_t L, int64_t Dn, int64_t k, int dtype_io,
                             void *stream);

/* ------------------------------------------------------------------------------------------
 * MoE router gating  (replaces AdaptiveExpertSystem.forward core.py:491-492,529)
 *   g = softmax(logits)              logits [S,E] fp32 (already noised if training)
 *   (p, idx) = topk(g, K) descending, ties -> lowest expert index first
 *   w = p / (sum(p) + 1e-6)
 * Outputs: gates [S,E] fp32, idx [S,K] int32, w [S,K] fp32.
 * Backward: given dw [S,K] and dgates [S,E] (from the aux losses) returns dlogits [S,E].
 * ------------------------------------------------------------------------------------------ */
int apertis_moe_gate_topk_fwd(const float *logits, float *gates, int32_t *idx, float *w,
                              int64_t S, int64_t E, int64_t K, void *stream);
int apertis_moe_gate_topk_bwd(const float *gates, const int32_t *idx, const float *dw,
                              const float *dgates, float *dlogits,
                              int64_t S, int64_t E, int64_t K, void *stream);

/* The same gate with the two router losses fused in (training, core.py:499-505, 524-526):
 *   stats[0] = lb_coef * E * sum_e (count_e / S) * mean_s gates[s,e]      (load balancing)
 *   stats[1] = rz_coef * mean_s logsumexp(logits[s,:])^2                  (router z-loss)
 *   stats[2..2+E) = count_e / S;  lse [S] = logsumexp per row (saved for the backward)
 * part = workspace [apertis_moe_gate_aux_blocks(S), 2E+1] fp32 (fixed-order fold).  Backward: dlb / drz are
 * DEVICE scalars (gradients of the two losses, NULL = 0); dw as above; returns dlogits. */
int apertis_moe_gate_topk_aux_fwd(const float *logits, float *gates, int32_t *idx, float *w,
                                  float *lse, float *part, float *stats, int64_t S, int64_t E,
                                  int64_t K, float lb_coef, float rz_coef, void *stream);
int apertis_moe_gate_topk_aux_bwd(const float *gates, const int32_t *idx, const float *dw,
                                  const float *lse, const float *stats, const float *dlb,
                                  const float *drz, float lb_coef, float rz_coef, float *dlogits,
                                  int64_t S, int64_t E, int64_t K, void *stream);
/* The same pair with the reference's NOISY top-k routing (core.py:485-488) inside the kernels:
 *   logits += n * softplus(w_noise[e]) * alpha,  n ~ N(0, 1)
 * with the normals drawn from a counter hash of (seed, token, expert pair) through Box-Muller, so
 * the backward regenerates them (same seed) instead of reading a [S, E] tensor.  w_noise [E] fp32
 * (NULL: no noise - what the plain entry points do).  Backward: npart = workspace
 * [apertis_moe_gate_aux_blocks(S)][E], dw_noise = out [E], the gradient of w_noise
 * (block sums of dlogits * n folded in a fixed order, times alpha * sigmoid(w_noise)). */
int apertis_moe_gate_topk_noisy_aux_fwd(const float *logits, const float *w_noise, float alpha,
                                        uint64_t seed, float *gates, int32_t *idx, float *w,
                                        float *lse, float *part, float *stats, int64_t S, int64_t E,
                                        int64_t K, float lb_coef, float rz_coef, void *stream);
int apertis_moe_gate_topk_noisy_aux_bwd(const float *gates, const int32_t *idx, const float *dw,
                                        const float *lse, const float *stats, const float *dlb,
                                        const float *drz, float lb_coef, float rz_coef,
                                        const float *w_noise, float alpha, uint64_t seed,
                                        float *dlogits, float *npart, float *dw_noise, int64_t S,
                                        int64_t E, int64_t K, void *stream);
int64_t apertis_moe_gate_aux_blocks(int64_t S);

/* Router projection y[T,N] = x[T,K] W[N,K]^T + b for N in {2,4,8,16}, K % 4 == 0, K <= 1024
 * (core.py:430,482: Linear(hidden -> num_experts)); fp32 weights/outputs, x fp32 or bf16.
 * Backward: dx [T,K] in x's dtype, dW_db = [N*K dW | N db] fp32; part = workspace
 * [apertis_skinny_linear_bwd_blocks(T), N*K+N] fp32 (deterministic fold). */
int apertis_skinny_linear_fwd(const void *x, const float *W, const float *b, float *y, int64_t T,
                              int64_t K, int64_t N, int dtype_x, void *stream);
int apertis_skinny_linear_bwd(const void *x, const float *W, const float *dy, void *dx,
                              float *part, float *dW_db, int64_t T, int64_t K, int64_t N,
                              int dtype_x, void *stream);
int64_t apertis_skinny_linear_bwd_blocks(int64_t T);

/* Router with its LayerNorm fused in (core.py:481-482): logits[T,N] = Linear(LayerNorm(x[T,H])),
 * N in {2,4,8}, H % 4 == 0, H <= 1024; x fp32 or bf16, everything else fp32; mean/rstd [T] saved.
 * Backward: dx [T,H] (x's dtype) = gradient through LN and the projection + dres (optional [T,H],
 * x's dtype: the gradient arriving on the other consumers of x, i.e. the expert path);
 * grads = [N*H dW | N db | H dgamma | H dbeta] fp32; part = workspace
 * [apertis_router_bwd_blocks(T), N*H + N + 2H] fp32 (fixed-order fold). */
int apertis_router_fwd(const void *x, const float *gamma, const float *beta, float eps,
                       const float *W, const float *b, float *logits, float *mean, float *rstd,
                       int64_t T, int64_t H, int64_t N, int dtype_x, void *stream);
int apertis_router_bwd(const void *x, const float *gamma, const float *beta, const float *mean,
                       const float *rstd, const float *W, const float *dlogits, const void *dres,
                       void *dx, float *part, float *grads, int64_t T, int64_t H, int64_t N,
                       int dtype_x, void *stream);
/* apertis_router_bwd with the expert path's gradient given as ROWS: row r additionally receives
 * round_x(sum_k grows[slot_of[r * KS + k]]) over its kept slots (slot_of < 0: none), k ascending - what
 * apertis_moe_combine_fwd(with_w = 0) would have put into a dense `dres`, without that tensor's round trip.
 * grows [rows, H] in x's dtype (the gather-LN backward's dxr), slot_of [T, KS] int32, KS <= 2.  dres may still be
 * given (both are added).  grows == NULL: apertis_router_bwd. */
int apertis_router_bwd_rows(const void *x, const float *gamma, const float *beta, const float *mean,
                            const float *rstd, const float *W, const float *dlogits, const void *dres,
                            const void *grows, const int32_t *slot_of, int64_t KS, void *dx, float *part,
                            float *grads, int64_t T, int64_t H, int64_t N, int dtype_x, void *stream);
int64_t apertis_router_bwd_blocks(int64_t T);
/* The router backward and the backward of the block boundary in front of it (apertis_dropout_add_layernorm_router_fwd's
 * two halves, core.py:481-482 behind :888,:847) in ONE pass over the rows: the total gradient of the normalised stream xn is
 * formed per row in registers - apertis_router_bwd_rows' arithmetic and rounding - and handed to apertis_layernorm_bwd's
 * arithmetic for the same row, so xn's gradient [T,H] is neither written nor read back.  dx (the residual stream's gradient:
 * LayerNorm backward + dres) and dblk (its masked copy, the block output's gradient) are bit-identical to the two calls;
 * the router's dW / db come from a second short launch, every affine gradient from a fixed-order fold.
 * y, dres, dx: dtype_x (fp32); xn, grows, dblk: dtype_g (bf16 / fp32).  No dense gradient term on xn (the caller uses the
 * two calls then).  part: workspace [apertis_router_bwd_blocks(T)][(N*H + N + 2H) + 2H] fp32;
 * rgrads: out [N*H dW | N db | H dgamma_r | H dbeta_r]; dgamma / dbeta [H]: the boundary norm's. */
int apertis_boundary_router_bwd(const void *y, const float *gamma, const float *mean, const float *rstd,
                                const void *dres, void *dx, void *dblk, float drop_p, uint64_t seed,
                                const void *xn, const float *rgamma, const float *rbeta, const float *rmean,
                                const float *rrstd, const float *W, const float *dlogits, const void *grows,
                                const int32_t *slot_of, int64_t KS, float *part, float *rgrads, float *dgamma,
                                float *dbeta, int64_t T, int64_t H, int64_t N, int dtype_x, int dtype_g,
                                void *stream);

/* Tiny linear y[T,N] = x[T,:K] W[N,K]^T + b for K <= 64, N <= 16: the SSM's
 * dt_proj_head (core.py:361,382) applied to a column slice of the x_param_proj output, read in
 * place (row stride ldx elements).  fp32 W/b/y, x fp32 or bf16.  Backward: dx rows (row stride lddx,
 * x's dtype), dW_db = [N*K dW | N db] fp32; part = workspace [apertis_tiny_linear_bwd_blocks(T),
 * N*K+N] fp32 (fixed-order fold). */
int apertis_tiny_linear_fwd(const void *x, int64_t ldx, const float *W, const float *b, float *y,
                            int64_t T, int64_t K, int64_t N, int dtype_x, void *stream);
int apertis_tiny_linear_bwd(const void *x, int64_t ldx, const float *W, const float *dy, void *dx,
                            int64_t lddx, float *part, float *dW_db, int64_t T, int64_t K,
                            int64_t N, int dtype_x, void *stream);
/* ... and with the columns [K, zero_to) of every dx row set to zero (zero_to <= lddx): the pad behind the dt columns in the
 * padded projection output's gradient, which otherwise costs a strided fill of its own. */
int apertis_tiny_linear_bwd_pad(const void *x, int64_t ldx, const float *W, const float *dy, void *dx,
                                int64_t lddx, float *part, float *dW_db, int64_t T, int64_t K,
                                int64_t N, int64_t zero_to, int dtype_x, void *stream);
int64_t apertis_tiny_linear_bwd_blocks(int64_t T);

/* out [B,N] = x W^T (+ bias [N] or NULL) for B <= 16 bf16 rows, K < 512, W [N, ldw] bf16 (K zero-padded to ldw): apertis_grouped_gemm_nt's
 * skinny kernel for one group with the row count by value (no load of group offsets in front of the operands) - the same bits. */
int apertis_decode_dense_gemv(const void *x, const void *W, int64_t ldw, const float *bias, void *out, int64_t B,
                              int64_t K, int64_t N, void *stream);
/* Single-token decode step, the in_proj product xz [S,N] = xn W^T (W [N, ldw] bf16; apertis_grouped_gemm_nt's skinny kernel for
 * K >= 512, the same bits; S <= 16, 512 <= H <= 1024) with
 *  - an optional PROLOGUE (xn == NULL): the block boundary in front of the SSM block - apertis_dropout_add_layernorm_fwd without
 *    dropout: y = res + blk (blk [S,H] bf16, or with slot_of / wk [S,KK] the MoE combine of yr rows taken on the fly), xn =
 *    LayerNorm(y) - which every work-group runs for itself (meant for S <= 4: a row per wave); y [S,H] fp32 is written;
 *  - an optional EPILOGUE (pre != NULL, N == 2 Dn; apertis_decode_post's arithmetic): the xp columns are pushed into conv_state
 *    [S,Dn,kconv-1] in place, gated [S,Dn] = pre * silu(z); xz itself is not written then (xz may be NULL). */
int apertis_decode_inproj(const void *blk, const int32_t *slot_of, const float *wk, int64_t KK, const float *res,
                          const float *gamma, const float *beta, float eps, float *y, const void *xn,
                          const void *W, int64_t ldw, void *xz, const float *pre, void *conv_state,
                          int64_t kconv, void *gated, int64_t S, int64_t H, int64_t N, int64_t Dn, void *stream);
/* The entrance of an MoE feed-forward for S <= 16 rows (the single-token decode step) in ONE launch:
 * apertis_dropout_add_layernorm_router_fwd without dropout (y = res + blk, xn = LayerNorm(y), logits = Linear(router_norm(xn)))
 * followed by apertis_moe_route_small on those logits and on xn - which stays in LDS (xn may be NULL).  The same arithmetic
 * and outputs as the two calls; E in {4, 8}, E*K <= 16, H <= 1024; res / y fp32 (dtype_x), blk / xn / xg dtype_y. */
int apertis_moe_enter_small(const void *blk, const void *res, const float *gamma, const float *beta,
                            float eps, void *y, void *xn, const float *rgamma, const float *rbeta,
                            float reps, const float *W, const float *rb, float *logits, float *gates,
                            int32_t *idx, float *w, int32_t *expert_offsets, int32_t *row_token,
                            int32_t *row_k, int32_t *slot_of, const float *lgamma, const float *lbeta,
                            float leps, void *xg, float *mean, float *rstd, int64_t S, int64_t H,
                            int64_t E, int64_t K, int dtype_x, int dtype_y, void *stream);
/* A handful of tokens (S <= 64, E*K <= 16: the single-token decode step, core.py:1578-1603): apertis_moe_gate_topk_fwd,
 * apertis_moe_plan (no capacity, no dropped experts) and apertis_moe_gather_ln_fwd as ONE launch of one work-group - the same
 * arithmetic, the same outputs (gates [S,E], idx / w [S,K]; the plan; xg [S*K,H] with mean / rstd per row). */
int apertis_moe_route_small(const float *logits, float *gates, int32_t *idx, float *w,
                            int32_t *expert_offsets, int32_t *row_token, int32_t *row_k,
                            int32_t *slot_of, const void *x, const float *gamma, const float *beta,
                            float eps, void *xg, float *mean, float *rstd, int64_t S, int64_t H,
                            int64_t E, int64_t K, int dtype_x, int dtype_xg, void *stream);
/* ------------------------------------------------------------------------------------------
 * MoE dispatch plan  (replaces the K x E Python loop core.py:547-591: nonzero / capacity /
 * overflow top-n by gate weight).  Canonical row order: expert-major, then k, then token
 * (equivalent to the reference's k-major loop because capacity is per expert).
 *   capacity <= 0 means "no limit" (eval, core.py:508).
 *   active   : optional [E] uint8 mask of experts not dropped (core.py:514-521), NULL = all.
 * Outputs:
 *   expert_offsets [E+1] int32 : row range of each expert in the permuted order
 *   row_token [S*K] int32, row_k [S*K] int32 : (token, k) of each kept row (first total rows)
 *   slot_of [S,K] int32 : permuted row of assignment (s,k) or -1 if dropped
 * Workspace ws: apertis_moe_plan_workspace_bytes(S,E,K) bytes, int32-aligned.
 * ------------------------------------------------------------------------------------------ */
int64_t apertis_moe_plan_workspace_bytes(int64_t S, int64_t E, int64_t K);
int apertis_moe_plan(const int32_t *idx, const float *w, const uint8_t *active,
                     int64_t capacity, int32_t *expert_offsets, int32_t *row_token,
                     int32_t *row_k, int32_t *slot_of, void *ws,
                     int64_t S, int64_t E, int64_t K, void *stream);

/* Gather + per-expert LayerNorm (core.py:593 gather, :436 expert LayerNorm):
 *   xg[r,:] = LN_e(x[row_token[r],:]) * gamma[e] + beta[e],   e = expert owning row r.
 * x [S,H] dtype_x; gamma,beta [E,H] fp32; xg [cap_rows,H] dtype_out; mean,rstd [cap_rows] fp32.
 * Rows >= expert_offsets[E] are left untouched.  max_rows bounds the launch (no host sync). */
int apertis_moe_gather_ln_fwd(const void *x, const int32_t *row_token,
                              const int32_t *expert_offsets, const float *gamma,
                              const float *beta, float eps, void *xg, float *mean, float *rstd,
                              int64_t max_rows, int64_t H, int64_t E, int dtype_x, int dtype_out,
                              void *stream);
/* Backward: dxg [rows,H] -> per-row dx contribution dxr [rows,H] (LayerNorm backward; dxr may be
 * NULL); the affine gradients are ADDED into dgamma/dbeta [E,H] fp32 (caller zero-fills): 32-row
 * blocks that lie inside one expert go through per-block partial sums (part, blk_expert: workspaces of
 * apertis_moe_gather_ln_bwd_blocks(max_rows) x 2H floats / ints) folded per expert in block order;
 * only blocks straddling an expert boundary use float atomics.  The rows are later scattered to
 * tokens by apertis_moe_combine_fwd(with_weights=0). */
int apertis_moe_gather_ln_bwd(const void *x, const int32_t *row_token,
                              const int32_t *expert_offsets, const float *gamma,
                              const float *mean, const float *rstd, const void *dxg,
                              void *dxr, float *dgamma, float *dbeta, float *part, int32_t *blk_expert,
                              int64_t max_rows, int64_t H, int64_t E, int dtype_x, int dtype_g,
                              void *stream);
int64_t apertis_moe_gather_ln_bwd_blocks(int64_t max_rows);

/* Plain LayerNorm over the last dimension on the same row kernels (pre-norms and final norm,
 * core.py:669,695,847,888,1040,1294): x [T,H] dtype_x -> y [T,H] dtype_y, mean/rstd [T] fp32.
 * Backward: dy [T,H] dtype_g -> dx [T,H] in dtype_x; dgamma/dbeta [H] fp32 overwritten;
 * part = workspace [apertis_layernorm_bwd_blocks(T,H), 2, H] fp32 (deterministic fold).
 * dres (optional, [T,H] dtype_x): gradient arriving on the residual branch around the norm
 * (pre-norm block y = x + f(LN(x)), core.py:667-700,845-890); dx = LN backward + dres.
 * dblk (optional, [T,H] dtype_g) with (drop_p, seed): when x itself was res + dropout(blk) - made by
 * apertis_dropout_add_layernorm_fwd - also writes blk's gradient, the masked copy of dx. */
int apertis_layernorm_fwd(const void *x, const float *gamma, const float *beta, float eps,
                          void *y, float *mean, float *rstd, int64_t T, int64_t H, int dtype_x,
                          int dtype_y, void *stream);
int apertis_layernorm_bwd(const void *x, const float *gamma, const float *mean,
                          const float *rstd, const void *dy, const void *dres, void *dx,
                          void *dblk, float drop_p, uint64_t seed, float *part,
                          float *dgamma, float *dbeta, int64_t T, int64_t H, int dtype_x,
                          int dtype_g, void *stream);
/* apertis_layernorm_bwd for a boundary whose block output was the MoE combine (apertis_dropout_add_layernorm_fwd called with
 * slot_of / wk: core.py:605 index_add_ of the weighted expert rows, then core.py:698 residual + dropout), with
 * apertis_moe_combine_bwd folded in: the masked gradient row [T, H] in the compute dtype is neither written nor read back.
 * dx, dgamma, dbeta as apertis_layernorm_bwd leaves them; dyr [rows, H] and dwk [T, K] (zeroed by the caller: dropped slots
 * are not written) as apertis_moe_combine_bwd would have made them, bit for bit.  K <= 2, else APERTIS_ERR_UNSUPPORTED:
 * call the two entry points.  `part` as for apertis_layernorm_bwd. */
int apertis_layernorm_combine_bwd(const void *x, const float *gamma, const float *mean, const float *rstd, const void *dy,
                                  const void *dres, void *dx, float drop_p, uint64_t seed, float *part, float *dgamma,
                                  float *dbeta, const int32_t *slot_of, const float *wk, const void *yr, void *dyr, float *dwk,
                                  int64_t T, int64_t H, int64_t K, int dtype_x, int dtype_g, void *stream);
int64_t apertis_layernorm_bwd_blocks(int64_t T, int64_t H);
/* Block boundary of the pre-norm stack in one pass: y = res + dropout(blk) (core.py:698,888; the mask of
 * apertis_dropout_add_fwd) and xn = LayerNorm(y) (the next sub-block's pre-norm).  res, y: dtype_x;
 * blk, xn: dtype_y.  With slot_of [T,K] / wk [T,K] (else NULL) blk is the MoE expert output [rows,H] and the
 * token's row is the combine sum_k wk*blk[slot_of] (apertis_moe_combine_fwd's arithmetic) taken on the fly.
 * Backward: apertis_layernorm_bwd with dblk (then apertis_moe_combine_bwd on dblk in the MoE form). */
int apertis_dropout_add_layernorm_fwd(const void *blk, const int32_t *slot_of, const float *wk, int64_t K,
                                      const void *res, const float *gamma,
                                      const float *beta, float eps, void *y, void *xn, float *mean,
                                      float *rstd, int64_t T, int64_t H, float drop_p, uint64_t seed,
                                      int dtype_x, int dtype_y, void *stream);
/* The same boundary in front of an MoE feed-forward, with the router projection of apertis_router_fwd in the same
 * pass: logits [T, N] = Linear(LayerNorm_r(xn)) (rgamma / rbeta / reps: the router's norm; W [N, H], rb [N] or NULL),
 * rmean / rrstd [T] = the router norm's statistics (what apertis_router_bwd wants).  blk is a dense [T, H] tensor here.
 * Bit-identical to apertis_dropout_add_layernorm_fwd followed by apertis_router_fwd on xn; N in {2, 4, 8}, H <= 1024. */
int apertis_dropout_add_layernorm_router_fwd(const void *blk, const void *res, const float *gamma,
                                             const float *beta, float eps, void *y, void *xn, float *mean,
                                             float *rstd, const float *rgamma, const float *rbeta, float reps,
                                             const float *W, const float *rb, float *logits, float *rmean,
                                             float *rrstd, int64_t T, int64_t H, int64_t N, float drop_p,
                                             uint64_t seed, int dtype_x, int dtype_y, void *stream);

/* Combine (core.py:594,605 weights * expert_output, index_add_):
 *   out[s,:] = sum_{k asc, slot_of[s,k]>=0} wk[s,k] * yr[slot_of[s,k],:]   (zeros if none)
 * with_weights=0 uses weight 1 (used to scatter the LN-backward rows).  */
int apertis_moe_combine_fwd(const void *yr, const int32_t *slot_of, const float *wk,
                            void *out, int64_t S, int64_t H, int64_t K, int with_weights,
                            int dtype_yr, int dtype_out, void *stream);
/* Backward of combine: dyr[r,:] = wk[s,k]*dout[s,:]; dwk[s,k] = <dout[s,:], yr[r,:]> */
int apertis_moe_combine_bwd(const void *dout, const void *yr, const int32_t *row_token,
                            const int32_t *row_k, const int32_t *expert_offsets,
                            const float *wk, void *dyr, float *dwk, int64_t max_rows,
                            int64_t S, int64_t H, int64_t K, int64_t E, int dtype_dout,
                            int dtype_yr, void *stream);

/* ------------------------------------------------------------------------------------------
 * Grouped GEMM on CDNA4 MFMA  (replaces the per-expert nn.Linear calls core.py:437,440,596;
 * with E=1 also the patch-embed Conv2d-as-GEMM multimodal/module.py:35-40,102 and
 * vision_projection core.py:1035,1209).
 *
 * Rows of A/C are grouped by expert: group e owns rows [offsets[e], offsets[e+1]).
 *   NT  (forward, and dgrad against the transposed weight copy):
 *         C[r,n]  = act( sum_k A[r,k] * W[e,n,k] + bias[e,n] )              W [E,N,K]
 *   TN  (wgrad)   dW[e,m,n] = sum_{r in e} A[r,m] * Bm[r,n],  dbias[e,m] = sum_r A[r,m]
 *   (the data gradient dX = dY * W is an NT product against W^T [E,K,N]; the compute copies
 *    W and W^T are produced together by apertis_cast_transpose, so every operand is
 *    K-contiguous and no NN kernel exists)
 * dtype: APERTIS_BF16 -> bf16 operands, fp32 accumulate on v_mfma_f32_16x16x32_bf16;
 *        APERTIS_F32  -> fp32 operands on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain).
 * max_rows bounds the grid (offsets live on the device; no host sync).
 * ldw (NT): row pitch of W in elements, 0 = K.  With ldw >= K rounded up to 64 and the pad columns
 *   zero (apertis_cast_transpose writes them), the 256x256 bf16 kernels also take K % 64 != 0.
 * pre_act (optional, NT only): also stores the pre-activation (needed by the backward).
 * act_bwd_pre (optional, dgrad fusion; excludes bias/pre_act): C = (A*W^T) (.) keepmask/(1-p) (.)
 *   act'(act_bwd_pre), i.e. the data gradient w.r.t. the PRE-activation of the producing layer, with
 *   the same (act, drop_p, seed) that layer's forward used - apertis_act_dropout_bwd fused away.
 * dropout (NT only): if drop_p>0 the activation output is multiplied by a keep mask /(1-p)
 *   from a counter-based hash of (seed, row, col) so the backward can regenerate it
 *   (reference: nn.Dropout inside each expert, core.py:439).
 * ------------------------------------------------------------------------------------------ */
int apertis_grouped_gemm_nt(const void *A, const void *W, const float *bias,
                            const int32_t *offsets, void *C, void *pre_act,
                            const void *act_bwd_pre,
                            int64_t max_rows, int64_t N, int64_t K, int64_t ldw, int64_t E,
                            int act, float drop_p, uint64_t seed,
                            int dtype, int dtype_out, void *stream);
/* 1 when the APERTIS_ACT_SAVE_GRAD / APERTIS_ACT_MUL_SAVED forms are available for this problem, else 0. */
int apertis_grouped_gemm_nt_saves_grad(int64_t max_rows, int64_t N, int64_t K, int64_t ldw, int64_t E,
                                       int act, int dtype, int dtype_out);
/* The same with a dynamic tile queue for the persistent 256 x 256 kernel: tile_queue = APERTIS_NT_QUEUE_INTS caller-owned
 * int32 (per stream; the entry point zeroes them on `stream` in front of the launch): one counter per XCD.  After its
 * first tile a work-group then draws its tiles from its XCD's counter - the tiles the static walk gives that XCD, so the
 * L2 sharing of neighbouring tiles is kept - and steals from the other XCDs' once that is drained, so a work-group whose
 * CU was held by a concurrent kernel (an RCCL collective on the communication stream of the data-parallel step) does not
 * finish a full share alone.  tile_queue == NULL: static schedule (what apertis_grouped_gemm_nt does). */
#define APERTIS_NT_QUEUE_INTS 512
int apertis_grouped_gemm_nt_q(const void *A, const void *W, const float *bias,
                              const int32_t *offsets, void *C, void *pre_act,
                              const void *act_bwd_pre,
                              int64_t max_rows, int64_t N, int64_t K, int64_t ldw, int64_t E,
                              int act, float drop_p, uint64_t seed,
                              int dtype, int dtype_out, int32_t *tile_queue, void *stream);
/* TN workspace (bf16 only): the 256x256-tile kernel deals the CUs out to the (problem, group)
 * pairs and splits the tiles left after the full rounds along the rows; the partial tiles live
 * in a caller-owned scratch buffer `ws` (16-byte aligned, apertis_grouped_gemm_tn_workspace_bytes
 * bytes, contents don't care) and are summed in a fixed order.  (The last 256 KiB of `ws` are the
 * item counters of the _q entry points below.)  ws == NULL, a too-small buffer
 * or E * n_problems > #CUs select the 128x128-tile kernel, which needs none.  The choice is the caller's: the 256x256
 * kernel pays off from about 2048 rows per group (below that a 256-row-deep slice per CU does not amortise the tile
 * prologue / epilogue and the fold); the library applies no threshold of its own and reads no environment variable.
 * With a workspace the pair entry points run 256x352 / 352x256 tiles when both problems have a side that those tile
 * without zero columns where 256x256 tiles would waste >= 5 % (the 704-wide family: [2816, 704] and [704, 2816]); the
 * workspace size covers either kernel. */
int64_t apertis_grouped_gemm_tn_workspace_bytes(int64_t E, int n_problems);
/* >= 0 when apertis_grouped_gemm_tn takes a ONE-group (E = 1) bf16 [M, N] weight gradient - a dense layer's, core.py:366-397:
 * the reduction runs over all rows - on its wide-tile kernel, splitting the rows over the CUs and folding the slices itself
 * (pass the workspace of apertis_grouped_gemm_tn_workspace_bytes(1, 1)); -1 when the caller should cut the rows into
 * pseudo-groups (offsets every 1024-2048 rows, E = their count, dW = [E, M, N] partial sums) and fold them with
 * apertis_colsum_f32, as the narrow shapes (under about 240 000 output elements, or tiles mostly padding) still do. */
int apertis_grouped_gemm_tn_dense_variant(int64_t M, int64_t N);
int apertis_grouped_gemm_tn(const void *A, const void *Bm, const int32_t *offsets,
                            float *dW, float *dbias, int64_t max_rows, int64_t M, int64_t N,
                            int64_t E, void *ws, int64_t ws_bytes, int dtype, void *stream);
/* Two TN problems over the SAME row grouping in one launch (the two expert layers' weight
 * gradients): one schedule over both, so neither pays tile-count quantisation alone. */
int apertis_grouped_gemm_tn_pair(const void *A0, const void *B0, float *dW0, float *dbias0,
                                 int64_t M0, int64_t N0, const void *A1, const void *B1,
                                 float *dW1, float *dbias1, int64_t M1, int64_t N1,
                                 const int32_t *offsets, int64_t max_rows, int64_t E,
                                 void *ws, int64_t ws_bytes, int dtype, void *stream);
/* The same two entry points with the 256x256 kernel's ITEM QUEUE (item_queue != 0; needs `ws`): work-groups take a
 * group's tiles from per-group counters kept in the last 256 KiB of `ws` (zeroed on `stream` by the entry point)
 * instead of a static share each, so a work-group that starts late - its CU held by a concurrent kernel such as an
 * RCCL collective of the data-parallel step - does not leave a full share undone.  Measured with
 * tools/probes/hog_probe.hip (32 of 256 CUs held): 2400 us static, 1900 us queue, 1460 us alone; alone the queue
 * costs 3 %, hence the switch.  The result is bit-identical either way. */
int apertis_grouped_gemm_tn_q(const void *A, const void *Bm, const int32_t *offsets,
                              float *dW, float *dbias, int64_t max_rows, int64_t M, int64_t N,
                              int64_t E, void *ws, int64_t ws_bytes, int dtype, int item_queue, void *stream);
int apertis_grouped_gemm_tn_pair_q(const void *A0, const void *B0, float *dW0, float *dbias0,
                                   int64_t M0, int64_t N0, const void *A1, const void *B1,
                                   float *dW1, float *dbias1, int64_t M1, int64_t N1,
                                   const int32_t *offsets, int64_t max_rows, int64_t E,
                                   void *ws, int64_t ws_bytes, int dtype, int item_queue, void *stream);
/* Compute copies of fp32 master weights src [E,R,C]: dst [E,R,ld_dst] and/or dstT [E,C,ld_dstT]
 * in dtype_out (either may be NULL).  Replaces what torch.autocast does per nn.Linear call.
 * ld_dst in [C, C rounded up to 64], ld_dstT in [R, R rounded up to 64] (0 = unpadded); pad columns
 * are written as zero - the form apertis_grouped_gemm_nt wants for K % 64 != 0. */
int apertis_cast_transpose(const float *src, void *dst, void *dstT, int64_t E, int64_t R,
                           int64_t C, int64_t ld_dst, int64_t ld_dstT, int dtype_out, void *stream);
/* The compute copies of MANY weights in ONE launch (the training step prepares every layer's stacked / padded / cast /
 * transposed weights once, at its start: core.py:366-397,437-440 read them).  `table`: device array of n_entries records of
 * apertis_weight_prep_entry_bytes() = 64 bytes each, in order of `tile0`:
 *   { const float *src;      fp32 [R, C], contiguous, 16-byte aligned, C % 4 == 0
 *     bf16 *plain;           or NULL: row r of src -> row (rowmap ? rowmap[r] : r), pitch ld_plain (% 4 == 0), 16-byte aligned
 *     bf16 *tr;              or NULL: the transposed copy, column index mapped the same way, pitch ld_tr (% 4 == 0)
 *     const int32_t *rowmap; or NULL
 *     int32_t R, C, ld_plain, ld_tr, tiles_c (= ceil(C / 64)), tile0 (first 64 x 64 tile of this entry in the launch), 0, 0 }
 * total_tiles = sum of ceil(R/64) * ceil(C/64).  Pad rows / columns of the destinations are not written (zero them once).
 * The conversion is apertis_cast_transpose's: the copies are bit-identical to its outputs. */
int64_t apertis_weight_prep_entry_bytes(void);
int apertis_weight_prep(const void *table, int64_t n_entries, int64_t total_tiles, void *stream);
/* out[c] = sum_r in[r,c] over a row-major fp32 [rows, cols] matrix, fixed summation order
 * (folds split-K partial weight gradients and per-block partial sums deterministically). */
int apertis_colsum_f32(const float *in, float *out, int64_t rows, int64_t cols, void *stream);
/* Elementwise backward of act+dropout: dpre = dh * mask/(1-p) * act'(pre). In place allowed. */
int apertis_act_dropout_bwd(const void *dh, const void *pre_act, void *dpre,
                            const int32_t *offsets, int64_t max_rows, int64_t N, int64_t E,
                            int act, float drop_p, uint64_t seed, int dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * Next-token cross entropy straight on the LM head's logits  (reference core.py:1407-1416:
 * CrossEntropyLoss(ignore_index=-100) over logits[..., :-1, :] vs labels[..., 1:], fp32 math).
 * logits [B, L, V] row-major (bf16 or fp32, V % 8 == 0 / V % 4 == 0, 16-byte aligned); labels
 * int64 [B, label_stride]; the target of row (b, l) is labels[b, l + 1] for l < n_pos
 * (n_pos <= L, n_pos + 1 <= label_stride).  Rows with l >= n_pos or target == ignore_index carry
 * no loss; any other target must lie in [0, V).
 * fwd: lse[b*L + l] = log sum exp(row), row_loss = lse - row[target] (0 for rows without loss);
 *      the caller reduces row_loss (sum / number of targets).
 * bwd: dlogits = (softmax(row) - onehot(target)) * gscale[0] (device scalar = dLoss / number of
 *      targets), zero rows where there is no loss; dlogits has the logits' dtype and layout.
 * ------------------------------------------------------------------------------------------ */
int apertis_cross_entropy_fwd(const void *logits, const int64_t *labels, float *lse,
                              float *row_loss, int64_t B, int64_t L, int64_t V,
                              int64_t label_stride, int64_t n_pos, int64_t ignore_index, int dtype,
                              void *stream);
int apertis_cross_entropy_bwd(const void *logits, const int64_t *labels, const float *lse,
                              const float *gscale, void *dlogits, int64_t B, int64_t L, int64_t V,
                              int64_t label_stride, int64_t n_pos, int64_t ignore_index, int dtype,
                              void *stream);
/* Both passes in ONE launch, for a caller that knows gscale before the forward (the fused LM head + loss, core.py:1412-1450 as
 * one op): lse / row_loss as apertis_cross_entropy_fwd leaves them and dlogits (may be `logits` itself) as
 * apertis_cross_entropy_bwd does, bit for bit, with one read of the logits (a row is kept in its work-group's registers between
 * the sweeps).  APERTIS_ERR_UNSUPPORTED when a row does not fit (V > 32768 bf16 / 16384 fp32) or the alignment rules of the
 * two entry points fail: call those instead. */
int apertis_cross_entropy_fwd_bwd(const void *logits, const int64_t *labels, float *lse, float *row_loss,
                                  const float *gscale, void *dlogits, int64_t B, int64_t L, int64_t V,
                                  int64_t label_stride, int64_t n_pos, int64_t ignore_index, int dtype, void *stream);

/* ---- optimizer step of the trainer (reference src/training/pipeline.py:469-473 AdamW groups, :544-546
 * clip_grad_norm_ + optimizer.step).  A parameter group is described by a DEVICE table of apertis_opt_tensor records - fp32
 * parameter, gradient, first and second moment, element count - and two int32 device arrays that cut the tensors into
 * chunks of apertis_opt_chunk_elems() consecutive elements: chunk c covers elements [chunk_index[c]*chunk,
 * +chunk) of tensor chunk_tensor[c].  The caller builds them once (pointers change only when tensors are reallocated). */
typedef struct apertis_opt_tensor {
  float *p, *g, *m, *v;
  int64_t numel;
} apertis_opt_tensor;
int64_t apertis_opt_chunk_elems(void);
/* partials[c] = sum of g^2 over chunk c (fixed summation order). */
int apertis_grad_sumsq(const void *tensors, const int32_t *chunk_tensor, const int32_t *chunk_index,
                       int64_t n_chunks, float *partials, void *stream);
/* norm_coef[0] = sqrt(sum of partials[0..n)) summed in index order, norm_coef[1] = min(1, max_norm/(norm+1e-6)):
 * torch.nn.utils.clip_grad_norm_'s coefficient, left on the device.  `poison` (may be NULL): a device int32 that, when
 * non-zero, makes the norm NaN and the coefficient -1, which apertis_adamw_step reads as "skip this step" - callers pass
 * the single-pass scan's error word (byte 8 of its workspace), so a look-back time-out rejects the optimizer step
 * visibly, without a host sync and without overwriting the parameters. */
int apertis_clip_coef(const float *partials, int64_t n, float max_norm, float *norm_coef, const int32_t *poison,
                      void *stream);
/* AdamW (torch.optim.AdamW's rule, decoupled decay, bias correction for `step` >= 1) on every chunk, with the gradient
 * scaled by norm_coef[1] when norm_coef != NULL (a negative coefficient: nothing is updated).  p, m, v are updated in
 * place; g is left as it was.  The hyper-
 * parameters are doubles: the derived constants (1-beta, 1-lr*wd, lr/bias_correction) are formed in double and rounded
 * once, as the Python optimizer does. */
int apertis_adamw_step(const void *tensors, const int32_t *chunk_tensor, const int32_t *chunk_index,
                       int64_t n_chunks, double lr, double beta1, double beta2, double eps, double weight_decay,
                       int64_t step, const float *norm_coef, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* APERTIS_HIP_H */
