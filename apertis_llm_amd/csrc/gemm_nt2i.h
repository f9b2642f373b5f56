// Included by grouped_gemm.hip (inside its anonymous namespace, behind grouped_gemm_nt4r_k): the saved-gradient forward of the
// expert MLP (reference core.py:434-442: Linear -> GELU -> Dropout; the forward also leaves gelu'(pre) * mask / (1 - p) for the
// backward) with the epilogue of tile i INSIDE the K loop of tile i + 1.
//
// 256 x 128 NT kernel, persistent, ONE wave per SIMD (round 6).  What rounds 3-4 measured: the fc1
// forward's epilogue (GELU, GELU', mask hash, two conversions, two stores per element pair: ~20 VALU instructions per output
// element) is as long as its K loop, and a VALU wave beside an MFMA wave on one SIMD makes no progress - two work-groups per CU
// (nt2x) or K-loop / epilogue wave roles cannot hide it.  The SAME wave can: tools/probes/mfma_valu_samewave.hip
// (profiles/r6_probe_mfma_valu_samewave.log) - one wave per SIMD, 32 MFMAs + 192 VALU instructions of the epilogue's mix per
// trip - runs at 0.74 of the sum of the two alone: the MFMAs cost a third of their own time next to enough VALU work.
// Here: 4 waves (one per SIMD, up to 512 registers), wave tile 64 x 128 like nt2x (X on the MFMA A operand, W rows permuted,
// the lane holds eight consecutive output columns of sixteen rows), a SIX-slot ring of 32-deep stages (24 KiB each, five stages
// = 120 KiB in flight: with one work-group per CU the ring is all the latency cover there is) that runs through tile boundaries
// like nt4r's.  At a tile's end the accumulators become the bf16-rounded pre-activations (bias added: what the epilogue rounds
// to first anyway) in 64 registers `pk`; sub-step s < 16 of the NEXT tile then carries row piece s of that epilogue - the
// arithmetic of nt2x_epilogue<EPI_BOTH>, operation for operation: outputs bit-identical - cut into eight chunks, one behind each
// group of four MFMAs, between sched_barrier fences (hipcc's own schedule, sched_group_barrier pipelines included, put the 32
// MFMAs first and the ~160 VALU instructions behind them).  The stream ends with one drain iteration on an empty tile
// (zero-fill DMA, MFMAs on zeros) that carries the last epilogue.
// K % 32 == 0, K >= 512 (sixteen sub-steps to put the sixteen row pieces in); GELU + saved gradient only.
constexpr int NSLOT5 = 6, RING5I = NSLOT5 * SLOT3;

template <typename TO, bool DROP>
__global__ void __launch_bounds__(NT3)
grouped_gemm_nt2i_k(const bf16_t *__restrict__ X, const bf16_t *__restrict__ W, const float *__restrict__ bias,
                    const int32_t *__restrict__ offsets, TO *__restrict__ C, TO *__restrict__ C2, int N, int K, int ldw, int E,
                    int n_tiles, int total_tiles, float drop_p, uint64_t seed, int walk_g, int walk_nb) {
  typedef bf16x8 frag;
  typedef unsigned u4_t __attribute__((ext_vector_type(4)));
  static_assert(sizeof(TO) == 2, "16-byte pieces of 2-byte outputs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fg = lane >> 4;
  const int nk = K / 32;   // >= 16 (launcher)
  const int ldb = K * 2, ldwb = ldw * 2;
  const int G = gridDim.x;

  auto decode = [&](int v) -> Tile4 {
    Tile4 t; t.valid = 0; t.e = 0; t.rows_valid = 0; t.n0 = 0; t.cols_valid = 0; t.row0 = 0;
    const int tile = xcd_remap(v, total_tiles);
    int mt, ntile;
    tile_walk(tile, total_tiles / n_tiles, n_tiles, walk_g, walk_nb, mt, ntile);
    int accm = 0;
    for (int g = 0; g < E; ++g) {
      const int r0 = offsets[g], r1 = offsets[g + 1];
      const int nt = (r1 - r0 + BM3 - 1) / BM3;
      if (mt < accm + nt) {
        const int m0 = (mt - accm) * BM3;
        t.valid = 1; t.e = g; t.row0 = r0 + m0; t.rows_valid = min(BM3, r1 - r0 - m0);
        break;
      }
      accm += nt;
    }
    t.e = __builtin_amdgcn_readfirstlane(t.e);
    t.valid = __builtin_amdgcn_readfirstlane(t.valid);
    t.rows_valid = __builtin_amdgcn_readfirstlane(t.rows_valid);
    t.row0 = (int64_t)__builtin_amdgcn_readfirstlane((int)t.row0);
    t.n0 = ntile * BN3; t.cols_valid = min(BN3, N - t.n0);
    return t;
  };
  int vnext = blockIdx.x;
  auto next_valid = [&]() -> Tile4 {
    Tile4 t; t.valid = 0; t.e = 0; t.rows_valid = 0; t.n0 = 0; t.cols_valid = 0; t.row0 = 0;
    while (vnext < total_tiles) {
      t = decode(vnext);
      vnext += G;
      if (t.valid) break;
    }
    return t;
  };
  Tile4 cur = next_valid();
  if (!cur.valid) return;

  // ---- the fill pointer (nt4r's scheme): (tile, sub-step) of the next stage to issue; past the last tile EMPTY descriptors ----
  const int fsw = (4 - ((lane >> 4) & 3)) & 3;
  const uint32_t vx0 = (uint32_t)((wave * 64 + (lane >> 2)) * ldb + (((lane & 3) ^ fsw) << 4));
  const uint32_t vw0 = (uint32_t)((lane >> 2) * 8 * ldwb + (((lane & 3) ^ fsw) << 4));
  const uint32_t lds0 = lds_addr_of(smem);
  v4i fxrs, fwrs, fbrs;
  int fvalid = 1;
  // the fill pointer runs six stages ahead of the multiply: it leaves the current tile at the bottom of sub-step nk - 7 and the
  // next tile's bias piece goes out at the top of sub-step nk - 6
  const int wrap_s = nk - 7;
  auto set_fill = [&](const Tile4 &t) {
    fxrs = raw_buffer_rsrc(X + t.row0 * K, t.valid ? (uint32_t)t.rows_valid * (uint32_t)ldb : 0u);
    fwrs = raw_buffer_rsrc(W + ((int64_t)t.e * N + t.n0) * ldw, t.valid ? (uint32_t)t.cols_valid * (uint32_t)ldwb : 0u);
    if (bias) fbrs = raw_buffer_rsrc(bias + (int64_t)t.e * N + t.n0, t.valid ? (uint32_t)t.cols_valid * 4u : 0u);
    fvalid = t.valid;
  };
  set_fill(cur);
  // the tile's bias segment (128 floats) as one more piece in front of its stage 0, into this wave's own 512 bytes (nt4r)
  const uint32_t bias_lds = lds0 + RING5I + wave * 1024;
  auto issue_bias = [&]() { lds_dma16(fbrs, bias_lds, lane < 32 ? (uint32_t)lane * 16u : 0xfffffff0u); };
  // this wave's six pieces of the fill stage: 0..3 its X pieces (rows wave*64 + q*16 ..), 4..5 its W pieces 2w, 2w + 1.  They go
  // out ONE BY ONE behind the first six MFMA groups of a sub-step: a wave sits in each `buffer_load ... lds` until the CU's
  // address unit has taken it (60-185 cycles: MI355X_MICROARCH.md), and with one wave per SIMD nothing else issues meanwhile -
  // behind a group of four MFMAs the matrix pipe at least has 64 cycles of work queued.  (kb = the stage's K offset in bytes.)
  uint32_t kb = 0;
  const uint32_t vxq1 = vx0 + (uint32_t)(16 * ldb), vxq2 = vx0 + (uint32_t)(32 * ldb), vxq3 = vx0 + (uint32_t)(48 * ldb);
  const uint32_t vwq0 = vw0 + (uint32_t)((wave * 2) * ldwb), vwq1 = vw0 + (uint32_t)((wave * 2 + 1) * ldwb);
  const uint32_t ldsx = lds0 + (uint32_t)(wave * 4 * 1024), ldsw = lds0 + (uint32_t)(BM3 * ROWB3 + wave * 2 * 1024);
  auto issue_piece = [&](uint32_t slot_off, auto qc) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q == 0) lds_dma16s(fxrs, ldsx + slot_off, vx0, kb);
    else if constexpr (q == 1) lds_dma16s(fxrs, ldsx + slot_off + 1024u, vxq1, kb);
    else if constexpr (q == 2) lds_dma16s(fxrs, ldsx + slot_off + 2048u, vxq2, kb);
    else if constexpr (q == 3) lds_dma16s(fxrs, ldsx + slot_off + 3072u, vxq3, kb);
    else if constexpr (q == 4) lds_dma16s(fwrs, ldsw + slot_off, vwq0, kb);
    else lds_dma16s(fwrs, ldsw + slot_off + 1024u, vwq1, kb);
  };
  auto issue_stage = [&](uint32_t slot_off) {   // (the prologue: all six at once)
    issue_piece(slot_off, std::integral_constant<int, 0>{}); issue_piece(slot_off, std::integral_constant<int, 1>{});
    issue_piece(slot_off, std::integral_constant<int, 2>{}); issue_piece(slot_off, std::integral_constant<int, 3>{});
    issue_piece(slot_off, std::integral_constant<int, 4>{}); issue_piece(slot_off, std::integral_constant<int, 5>{});
  };
  Tile4 nxt;
  nxt.valid = 0; nxt.e = 0; nxt.rows_valid = 0; nxt.n0 = 0; nxt.cols_valid = 0; nxt.row0 = 0;

  f32x4 acc[4][8];
  const int frd = frow * ROWB3 + ((fg ^ ((4 - ((frow >> 2) & 3)) & 3)) << 4);
  const char *abase = smem + wave * 64 * ROWB3 + frd, *bbase = smem + BM3 * ROWB3 + frd;
  frag af[2][4], bfr[8];

  // ---- the previous tile's epilogue state ----
  uint32_t pk[16][4];   // row piece R = i*4 + q: eight bf16 pre-activations (columns frow*8 .. +7 of tile row wave*64 + i*16 + fg*4 + q)
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int w = 0; w < 4; ++w) pk[r][w] = 0u;
  const float keep_scale = DROP ? 1.f / (1.f - drop_p) : 1.f;
  const uint32_t thresh16 = (uint32_t)(drop_p * 65536.f);
  const GeluK gk = gelu_consts(keep_scale);
  __amdgpu_buffer_rsrc_t o1 = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0, 0x00020000), o2 = o1;   // (empty: the first iteration's stores are dropped)
  uint32_t pvoff = 0xC0000000u, prow_base = 0u;
  uint32_t cmix[4] = {0u, 0u, 0u, 0u};
  const int rl = wave * 64 + fg * 4;
  auto set_prev = [&](const Tile4 &t) {   // the tile whose accumulators were just packed becomes the epilogue's tile
    const int64_t tile0 = t.row0 * N + t.n0;
    const uint32_t tile_bytes = t.valid ? (uint32_t)t.rows_valid * (uint32_t)N * 2u : 0u;
    o1 = __builtin_amdgcn_make_buffer_rsrc(C + tile0, 0, tile_bytes, 0x00020000);
    o2 = __builtin_amdgcn_make_buffer_rsrc(C2 + tile0, 0, tile_bytes, 0x00020000);
    pvoff = frow * 8 < t.cols_valid ? ((uint32_t)rl * (uint32_t)N + (uint32_t)(frow * 8)) * 2u : 0xC0000000u;
    prow_base = (uint32_t)((int)t.row0 + rl);
    if (DROP) {
#pragma unroll
      for (int w = 0; w < 4; ++w) cmix[w] = gd_colmix(seed, (uint32_t)((t.n0 + frow * 8) >> 1) + (uint32_t)w);
    }
  };
  // Row piece R of the previous tile - nt2x_epilogue<EPI_BOTH>'s arithmetic from the rounded pre-activation on (gelu_both_fast8's
  // stages, operation for operation) - in EIGHT chunks.  State that lives across the chunks:
  v2f ex[4], et[4], ee[4], ep[4], eq[4];
  uint32_t edm[4] = {0u, 0u, 0u, 0u};
  auto epi_chunk = [&](auto rc, auto cc) {
    constexpr int R = decltype(rc)::value, CH = decltype(cc)::value, r = (R >> 2) * 16 + (R & 3);
    if constexpr (CH == 0) {          // the row's mask words; the pre-activation pairs
      if (DROP) {
        const uint32_t rmix = gd_rowmix(seed, (uint64_t)(prow_base + (uint32_t)r));
#pragma unroll
        for (int w = 0; w < 4; ++w) edm[w] = gd_pair(rmix, cmix[w]);
      }
#pragma unroll
      for (int w = 0; w < 4; ++w)
        ex[w] = (v2f){__builtin_bit_cast(float, pk[R][w] << 16), __builtin_bit_cast(float, pk[R][w] & 0xffff0000u)};
    } else if constexpr (CH == 1) {   // 1 + k |x|, x^2 * (-log2(e) / 2); the keep masks
      const float tk = GELU_TK;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        asm("v_fma_f32 %0, |%1|, %2, 1.0" : "=v"(et[w].x) : "v"(ex[w].x), "s"(tk));
        asm("v_fma_f32 %0, |%1|, %2, 1.0" : "=v"(et[w].y) : "v"(ex[w].y), "s"(tk));
      }
#pragma unroll
      for (int w = 0; w < 4; ++w) ee[w] = (ex[w] * ex[w]) * splat2(-0.5f * LOG2E_F);
      if (DROP) {
#pragma unroll
        for (int w = 0; w < 4; ++w) edm[w] = drop_mask2(edm[w], thresh16);
      }
    } else if constexpr (CH == 2) {   // the reciprocals
#pragma unroll
      for (int w = 0; w < 4; ++w) et[w] = (v2f){__builtin_amdgcn_rcpf(et[w].x), __builtin_amdgcn_rcpf(et[w].y)};
    } else if constexpr (CH == 3) {   // the exponentials
#pragma unroll
      for (int w = 0; w < 4; ++w) ee[w] = (v2f){__builtin_amdgcn_exp2f(ee[w].x), __builtin_amdgcn_exp2f(ee[w].y)};
    } else if constexpr (CH == 4) {   // the erfc polynomial
#pragma unroll
      for (int w = 0; w < 4; ++w) ep[w] = pk_fma(splat2(gk.c3), et[w], splat2(gk.c2));
#pragma unroll
      for (int w = 0; w < 4; ++w) ep[w] = pk_fma(ep[w], et[w], splat2(gk.c1));
#pragma unroll
      for (int w = 0; w < 4; ++w) eq[w] = ep[w] * et[w];
#pragma unroll
      for (int w = 0; w < 4; ++w) eq[w] = eq[w] * ee[w];
    } else if constexpr (CH == 5) {   // P = s Phi(x); x s / sqrt(2 pi)
#pragma unroll
      for (int w = 0; w < 4; ++w) ep[w] = splat2(gk.s) - eq[w];
#pragma unroll
      for (int w = 0; w < 4; ++w) ep[w] = (v2f){ex[w].x >= 0.f ? ep[w].x : eq[w].x, ex[w].y >= 0.f ? ep[w].y : eq[w].y};
#pragma unroll
      for (int w = 0; w < 4; ++w) et[w] = ex[w] * splat2(gk.sphi);
    } else if constexpr (CH == 6) {   // gelu' = P + (x sphi) e (into ee), gelu = x P (into ep)
#pragma unroll
      for (int w = 0; w < 4; ++w) ee[w] = pk_fma(et[w], ee[w], ep[w]);
#pragma unroll
      for (int w = 0; w < 4; ++w) ep[w] = ex[w] * ep[w];
    } else {                          // conversions, masks, the two stores
      const uint32_t off = pvoff + (uint32_t)(r * N) * 2u;
      uint32_t o[4], og[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        o[w] = pack_bf16x2(ep[w].x, ep[w].y) & ~edm[w];
        og[w] = pack_bf16x2(ee[w].x, ee[w].y) & ~edm[w];
      }
      __builtin_amdgcn_raw_buffer_store_b128((u4_t){o[0], o[1], o[2], o[3]}, o1, (int)off, 0, 2);
      __builtin_amdgcn_raw_buffer_store_b128((u4_t){og[0], og[1], og[2], og[3]}, o2, (int)off, 0, 2);
    }
  };

  // ---- one group of four MFMAs on (acur, bfr[j]) with the next sub-step's fragments read under them (nt2x's scheme: W in
  // place once its four MFMAs have issued, X into the other set) ----
  auto mfma_group = [&](auto jc, const frag (&acur)[4], frag (&anxt)[4], int nxt_off) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j == 0) { anxt[0] = *reinterpret_cast<const frag *>(abase + nxt_off); anxt[1] = *reinterpret_cast<const frag *>(abase + nxt_off + 16 * ROWB3); }
    if constexpr (j == 1) { anxt[2] = *reinterpret_cast<const frag *>(abase + nxt_off + 32 * ROWB3); anxt[3] = *reinterpret_cast<const frag *>(abase + nxt_off + 48 * ROWB3); }
#pragma unroll
    for (int i = 0; i < 4; ++i) mma(acc[i][j], acur[i], bfr[j]);
    bfr[j] = *reinterpret_cast<const frag *>(bbase + nxt_off + j * 16 * ROWB3);
  };
  int cur_off = 0;      // LDS offset of the slot of the sub-step about to be multiplied
  // the wait at the top of sub-step S: stage S + 1 must have landed; younger than it: the stages issued in the last four
  // sub-steps (24 pieces) and the epilogue stores issued there (two per sub-step S' < 16).  (The bias piece is not counted: the
  // waits are one operation stronger where it is among the younger ones.)
#define NT2I_TOP(S, NV)                                                                                 \
      const int nxt_off = cur_off + SLOT3 == RING5I ? 0 : cur_off + SLOT3;                              \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
      wait_vmcnt<NV>();                                                                                 \
      lds_barrier();                                                                                    \
      if ((S) == wrap_s + 1 && bias && fvalid) issue_bias();                                            \
      __builtin_amdgcn_sched_barrier(0);
#define NT2I_BOTTOM(S)                                                                                  \
      kb += ROWB3;                                                                                      \
      if ((S) == wrap_s) { kb = 0; set_fill(nxt); }                                                     \
      cur_off = nxt_off;
#define NT2I_IC(v_) std::integral_constant<int, v_>{}
#define NT2I_G(J, AC, AN) mfma_group(NT2I_IC(J), AC, AN, nxt_off); __builtin_amdgcn_sched_barrier(0);
#define NT2I_P(Q) issue_piece((uint32_t)cur_off, NT2I_IC(Q)); __builtin_amdgcn_sched_barrier(0);
#define NT2I_C(R, CH) epi_chunk(NT2I_IC(R), NT2I_IC(CH)); __builtin_amdgcn_sched_barrier(0);
#define NT2I_EPI(R, NV, AC, AN)                                                                         \
    {                                                                                                   \
      NT2I_TOP(R, NV)                                                                                   \
      NT2I_G(0, AC, AN) NT2I_P(0) NT2I_C(R, 0) NT2I_G(1, AC, AN) NT2I_P(1) NT2I_C(R, 1) NT2I_G(2, AC, AN) NT2I_P(2) NT2I_C(R, 2) \
      NT2I_G(3, AC, AN) NT2I_P(3) NT2I_C(R, 3) NT2I_G(4, AC, AN) NT2I_P(4) NT2I_C(R, 4) NT2I_G(5, AC, AN) NT2I_P(5) NT2I_C(R, 5) \
      NT2I_G(6, AC, AN) NT2I_C(R, 6) NT2I_G(7, AC, AN) NT2I_C(R, 7)                                     \
      NT2I_BOTTOM(R)                                                                                    \
    }
#define NT2I_PLAIN(S, NV, AC, AN)                                                                       \
    {                                                                                                   \
      NT2I_TOP(S, NV)                                                                                   \
      NT2I_G(0, AC, AN) NT2I_P(0) NT2I_G(1, AC, AN) NT2I_P(1) NT2I_G(2, AC, AN) NT2I_P(2) NT2I_G(3, AC, AN) NT2I_P(3) \
      NT2I_G(4, AC, AN) NT2I_P(4) NT2I_G(5, AC, AN) NT2I_P(5) NT2I_G(6, AC, AN) NT2I_G(7, AC, AN)       \
      NT2I_BOTTOM(S)                                                                                    \
    }

  // prologue: the first six stages of the stream (slots 0..5; nk >= 16: all of cur).  Stage t lives in slot t % 6; the
  // fragments of sub-step s are read during s - 1 (nt2x / nt4r), so slot(s) is free behind the barrier at the top of s and
  // stage s + 6 goes there: stages s + 2 .. s + 6 are in flight under the MFMAs of s, stage s + 1 is being read.
  if (bias) issue_bias();
#pragma unroll 1
  for (int k = 0; k < NSLOT5; ++k) { issue_stage((uint32_t)(k * SLOT3)); kb += ROWB3; }
  wait_vmcnt<30>();   // stage 0 (and the bias piece in front of it)
  lds_barrier();
  for (;;) {
    nxt = cur.valid ? next_valid() : cur;   // (cur invalid: the drain iteration - nothing follows)
    // stage 0's fragments (read again here rather than held across the previous tile's tail; the slot is refilled only behind
    // sub-step 0's barrier), the tile's bias
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const frag *>(abase + cur_off + i * 16 * ROWB3);
#pragma unroll
    for (int j = 0; j < 8; ++j) bfr[j] = *reinterpret_cast<const frag *>(bbase + cur_off + j * 16 * ROWB3);
    float bv[8];
    {
      const float4 b0 = bias ? *reinterpret_cast<const float4 *>(smem + RING5I + wave * 1024 + frow * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 b1 = bias ? *reinterpret_cast<const float4 *>(smem + RING5I + wave * 1024 + frow * 32 + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
      bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // sub-steps 0..15 carry the previous tile's sixteen row pieces; stores behind: two per earlier sub-step of this tile, <= 4 back
    NT2I_EPI(0, 24, af[0], af[1])  NT2I_EPI(1, 26, af[1], af[0])  NT2I_EPI(2, 28, af[0], af[1])  NT2I_EPI(3, 30, af[1], af[0])
    NT2I_EPI(4, 32, af[0], af[1])  NT2I_EPI(5, 32, af[1], af[0])  NT2I_EPI(6, 32, af[0], af[1])  NT2I_EPI(7, 32, af[1], af[0])
    NT2I_EPI(8, 32, af[0], af[1])  NT2I_EPI(9, 32, af[1], af[0])  NT2I_EPI(10, 32, af[0], af[1]) NT2I_EPI(11, 32, af[1], af[0])
    NT2I_EPI(12, 32, af[0], af[1]) NT2I_EPI(13, 32, af[1], af[0]) NT2I_EPI(14, 32, af[0], af[1]) NT2I_EPI(15, 32, af[1], af[0])
    for (int s = 16; s < nk; s += 2) {
      if (s == 16) NT2I_PLAIN(s, 32, af[0], af[1]) else if (s == 18) NT2I_PLAIN(s, 28, af[0], af[1]) else NT2I_PLAIN(s, 24, af[0], af[1])
      if (s + 1 < nk) {
        if (s == 16) NT2I_PLAIN(s + 1, 30, af[1], af[0]) else if (s == 18) NT2I_PLAIN(s + 1, 26, af[1], af[0]) else NT2I_PLAIN(s + 1, 24, af[1], af[0])
      } else {   // odd nk: keep the register roles of the loop
#pragma unroll
        for (int i = 0; i < 4; ++i) af[0][i] = af[1][i];
      }
    }
    // the tile's accumulators -> the rounded pre-activations the epilogue starts from (nt2x_epilogue<EPI_BOTH>: one v_add_f32 per
    // element, then a packed conversion per pair)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) asm("v_add_f32 %0, %1, %2" : "=v"(v[j]) : "v"(acc[i][j][q]), "v"(bv[j]));
#pragma unroll
        for (int w = 0; w < 4; ++w) pk[i * 4 + q][w] = pack_bf16x2(v[2 * w], v[2 * w + 1]);
      }
    set_prev(cur);
    if (!cur.valid) break;       // (that was the drain iteration: its "tile" was empty and the last epilogue has been issued)
    cur = nxt;                   // (invalid past the last tile: one more iteration for the epilogue of the tile just packed)
  }
#undef NT2I_EPI
#undef NT2I_PLAIN
#undef NT2I_TOP
#undef NT2I_BOTTOM
#undef NT2I_G
#undef NT2I_C
#undef NT2I_P
#undef NT2I_IC
  wait_vmcnt<0>();
}
