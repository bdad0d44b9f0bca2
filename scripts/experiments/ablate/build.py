"""Developer tool: builds ablated variants of the engine (pieces of the adjoint kernel compiled out --
the RESULTS of these libraries are wrong, only their timing means something) into
scripts/ablate/lib_<name>.so; time them on the GPU with
    for f in scripts/ablate/lib_*.so; do QHBM_ENGINE_LIB=$f python scripts/vqt_time.py 512; done
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CSRC = os.path.join(ROOT, "qhbm-library_amd", "csrc")
OUT = os.path.dirname(os.path.abspath(__file__))


def kernel_source():
  with open(os.path.join(CSRC, "kernels.hip")) as f:
    return f.read()


def once(text, old, new):
  assert text.count(old) >= 1, old
  return text.replace(old, new, 1)


def inst(text):  # the body of instance_adj
  a = text.index("__device__ __forceinline__ void instance_adj(")
  b = text.index("// Writes the tile's gradient row")
  return a, b


def in_instance(text, old, new, count=-1):
  a, b = inst(text)
  body = text[a:b]
  assert old in body, old
  return text[:a] + body.replace(old, new, count) + text[b:]


VARIANTS = {
    "base": lambda t: t,
    "no_reduce": lambda t: once(t, "  float t0, t1, t2, t3, u0, u1, w, x;\n",
                                "  if (sv == 0x12345u) return;\n  if (true) return;\n  float t0, t1, t2, t3, u0, u1, w, x;\n"),
    # the reductions' cross-lane part compiled out: the partial sums stay alive (an empty asm consumes them), one of
    # them is stored -- what is removed is add_slots8's DPP adds, swaps and tail
    "no_butterfly": lambda t: once(t, "  float t0, t1, t2, t3, u0, u1, w, x;\n",
                                   "  if (lane >= 0) { asm volatile(\"\" :: \"v\"(g0), \"v\"(g1), \"v\"(g2), \"v\"(g3), \"v\"(g4), \"v\"(g5), \"v\"(g6), \"v\"(g7), \"s\"(present)); if ((lane & 3) == (G8 & 3) && (lane >> 5) == (G8 >> 2) && sv != 0xffffffffu) cells[sv * NW + wave] = g0; return; }\n  float t0, t1, t2, t3, u0, u1, w, x;\n"),
    # record coefficients as compile-time constants (only the two header words are loaded): what the
    # scalar-load latency of the record fields costs
    "const_coefs": lambda t: once(t, "  if constexpr (QHBM_SCALAR_RECORDS) return rb.p[W];", "  if constexpr (W >= 2) return 0x3f19999au; else if constexpr (QHBM_SCALAR_RECORDS) return rb.p[W];"),
    # wave priority: instances at high priority, exchanges / tile I/O at low
    "setprio": lambda t: once(once(t, "        instance_adj<R, NW, false>(cur, sv, recs, rec_off, lane, wave, p, l, TL | tile_hi, cells);\n        rec_off += L.words();\n        cur[0] = nxt[0];",
                                      "        __builtin_amdgcn_s_setprio(3);\n        instance_adj<R, NW, false>(cur, sv, recs, rec_off, lane, wave, p, l, TL | tile_hi, cells);\n        __builtin_amdgcn_s_setprio(0);\n        rec_off += L.words();\n        cur[0] = nxt[0];"),
                              "        instance_fwd<R, NV, GEN>(cur, recs, rec_off, lane, amp, TL | tile_hi);",
                              "        __builtin_amdgcn_s_setprio(3);\n        instance_fwd<R, NV, GEN>(cur, recs, rec_off, lane, amp, TL | tile_hi);\n        __builtin_amdgcn_s_setprio(0);"),
    "setprio_inv": lambda t: once(once(t, "        instance_adj<R, NW, false>(cur, sv, recs, rec_off, lane, wave, p, l, TL | tile_hi, cells);\n        rec_off += L.words();\n        cur[0] = nxt[0];",
                                      "        __builtin_amdgcn_s_setprio(0);\n        instance_adj<R, NW, false>(cur, sv, recs, rec_off, lane, wave, p, l, TL | tile_hi, cells);\n        __builtin_amdgcn_s_setprio(3);\n        rec_off += L.words();\n        cur[0] = nxt[0];"),
                              "        instance_fwd<R, NV, GEN>(cur, recs, rec_off, lane, amp, TL | tile_hi);",
                              "        __builtin_amdgcn_s_setprio(0);\n        instance_fwd<R, NV, GEN>(cur, recs, rec_off, lane, amp, TL | tile_hi);\n        __builtin_amdgcn_s_setprio(3);"),
    # adjoint without the LDS staging of the tile pair at the start and the end of a pass (registers filled
    # from / stored to HBM in the prefetch layout -- wrong amplitudes, same traffic): the upper bound of
    # loading and storing in the first / last round's geometry directly
    "adj_no_staging": lambda t: once(once(once(t,
        "template <int K, int ROWS>\n__global__ __launch_bounds__(1 << (K - 4), adjx_min_waves(K)) void pass_adjx_kernel(",
        """__device__ __forceinline__ void regs_from_tile(v2f (&a)[16], const TileRegs& r) {
  a[0] = v2f{r.p0.x, r.p0.y}; a[1] = v2f{r.p0.z, r.p0.w}; a[2] = v2f{r.p1.x, r.p1.y}; a[3] = v2f{r.p1.z, r.p1.w};
  a[4] = v2f{r.p2.x, r.p2.y}; a[5] = v2f{r.p2.z, r.p2.w}; a[6] = v2f{r.p3.x, r.p3.y}; a[7] = v2f{r.p3.z, r.p3.w};
  a[8] = v2f{r.p4.x, r.p4.y}; a[9] = v2f{r.p4.z, r.p4.w}; a[10] = v2f{r.p5.x, r.p5.y}; a[11] = v2f{r.p5.z, r.p5.w};
  a[12] = v2f{r.p6.x, r.p6.y}; a[13] = v2f{r.p6.z, r.p6.w}; a[14] = v2f{r.p7.x, r.p7.y}; a[15] = v2f{r.p7.z, r.p7.w};
}
template <int K, int NT>
__device__ __forceinline__ void regs_to_global(const v2f (&a)[16], float2* __restrict__ st, const TileCtx& t, int tid) {
  const uint32_t g0 = tile_offset(t, 2u * uint32_t(tid));
#define QHBM_RG(I) { float2* sb = st + (t.tile_base | t.ro[I]); *reinterpret_cast<float4*>(sb + g0) = make_float4(a[2 * I].x, a[2 * I].y, a[2 * I + 1].x, a[2 * I + 1].y); }
  QHBM_RG(0) QHBM_RG(1) QHBM_RG(2) QHBM_RG(3) QHBM_RG(4) QHBM_RG(5) QHBM_RG(6) QHBM_RG(7)
#undef QHBM_RG
}
template <int K, int ROWS>
__global__ __launch_bounds__(1 << (K - 4), adjx_min_waves(K)) void pass_adjx_kernel("""),
        "  commit_tile<K, NT>(xt, rp, tid);\n  __syncthreads();\n  round_load0<R>(T, DB, p);\n  __syncthreads();\n  commit_tile<K, NT>(xt, rl, tid);\n  __syncthreads();\n  round_load0<R>(T, DB, l);\n",
        "  regs_from_tile(p, rp);\n  regs_from_tile(l, rl);\n  __syncthreads();\n"),
        "    const ThreadOff o = thread_offsets<ROWS>(t, tid);\n    round_store0<R>(T, DB, p);\n    __syncthreads();\n    __builtin_amdgcn_sched_barrier(0);\n    store_tile<K, NT, ROWS>(xt, sp, t, o, tid);\n    __builtin_amdgcn_sched_barrier(0);\n    __syncthreads();\n    round_store0<R>(T, DB, l);\n    __syncthreads();\n    __builtin_amdgcn_sched_barrier(0);\n    store_tile<K, NT, ROWS>(xt, sl, t, o, tid);\n",
        "    regs_to_global<K, NT>(p, sp, t, tid);\n    regs_to_global<K, NT>(l, sl, t, tid);\n    __syncthreads();\n"),
    "no_x_inner": lambda t: in_instance(t, "g[J] = im_lam_x_psi<R, J>(p, l);", "g[J] = p[0].x;"),
    "no_x_on_lambda": lambda t: in_instance(t, "          apply_x<R, J>(l, cs);\n", ""),
    "no_x_at_all": lambda t: in_instance(in_instance(in_instance(t, "          apply_x<R, J>(l, cs);\n", ""), "          apply_x<R, J>(p, cs);\n", ""),
                                         "g[J] = im_lam_x_psi<R, J>(p, l);", "g[J] = p[0].x + cs.x;"),
    "no_full": lambda t: in_instance(t, "  if (h1 & kFullDiagFlag) {", "  if ((h1 & kFullDiagFlag) && lane == 77) {"),
    "no_cph": lambda t: in_instance(t, "  if (h1 & 0xffu) {", "  if ((h1 & 0xffu) && lane == 77) {", 1),
    "no_ph1_ph2": lambda t: in_instance(in_instance(t, "  if ((h0 >> 16) & 0x3fu) {", "  if (((h0 >> 16) & 0x3fu) && lane == 77) {"),
                                        "  if ((h0 >> 8) & 0xfu) {", "  if (((h0 >> 8) & 0xfu) && lane == 77) {"),
    "no_exchange": lambda t: once(t, """    round_store0<R>(T, DB, p);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, p);
    if (sync) __syncthreads();
    round_store0<R>(T, DB, l);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, l);
""", ""),
    "no_barriers": lambda t: once(t, """    round_store0<R>(T, DB, p);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, p);
    if (sync) __syncthreads();
    round_store0<R>(T, DB, l);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, l);
""", """    round_store0<R>(T, DB, p);
    round_load0<R>(Tn, DBn, p);
    round_store0<R>(T, DB, l);
    round_load0<R>(Tn, DBn, l);
"""),
    "no_instances": lambda t: once(t, "        instance_adj<R, NW, false>(cur, sv, recs, rec_off, lane, wave, p, l, TL | tile_hi, cells);\n        rec_off += L.words();\n        cur[0] = nxt[0];",
                                   "        if (lane == 77) instance_adj<R, NW, false>(cur, sv, recs, rec_off, lane, wave, p, l, TL | tile_hi, cells);\n        rec_off += L.words();\n        cur[0] = nxt[0];"),
}

def in_fwd_instance(text, old, new, count=-1):
  a = text.index("__device__ __forceinline__ void instance_fwd(")
  b = text.index("// Measurement helpers")
  body = text[a:b]
  assert old in body, old
  return text[:a] + body.replace(old, new, count) + text[b:]


VARIANTS.update({
    "fwd_no_instances": lambda t: once(t, "        instance_fwd<R, NV, GEN>(cur, recs, rec_off, lane, amp, TL | tile_hi);",
                                       "        if (lane == 77) instance_fwd<R, NV, GEN>(cur, recs, rec_off, lane, amp, TL | tile_hi);"),
    "fwd_no_x": lambda t: in_fwd_instance(t, "if ((h0 >> J) & 1u) apply_x<R, J>(a, rec_cs<L.x(J)>(rv, rb));", "if (((h0 >> J) & 1u) && lane == 77) apply_x<R, J>(a, rec_cs<L.x(J)>(rv, rb));"),
    "fwd_no_full": lambda t: in_fwd_instance(t, "if (h1 & kFullDiagFlag) apply_full<NV>(a, rv, rb, false);", "if ((h1 & kFullDiagFlag) && lane == 77) apply_full<NV>(a, rv, rb, false);"),
    "fwd_no_cph": lambda t: in_fwd_instance(t, "  if (h1 & 0xffu) {", "  if ((h1 & 0xffu) && lane == 77) {"),
    # forward without its HBM traffic (tile neither loaded nor stored): what the memory phase adds to the compute
    "fwd_no_tile_io": lambda t: once(once(t, "    prefetch_tile<K, NT>(r, ld, t, toff);\n    if (a.frozen_old_local) clear_stale",
                                          "    r = TileRegs{};\n    if (a.frozen_old_local) clear_stale"),
                                     "  if (a.flags & PASS_STORE) store_tile<K, NT>(tile, st, t, thread_offsets(t, tid), tid);\n}\n", "  if ((a.flags & PASS_STORE) && tid == 1000) store_tile<K, NT>(tile, st, t, thread_offsets(t, tid), tid);\n}\n"),
    "fwd_const_no_trips": lambda t: VARIANTS["const_coefs"](VARIANTS["fwd_no_round_trips"](t)),
    # stagger the first generation of forward workgroups (by hardware wave slot) to break lock step
    "fwd_stagger": lambda t: once(t, "    prefetch_tile<K, NT>(r, ld, t, toff);\n    if (a.frozen_old_local) clear_stale",
                                  "    if (blockIdx.x < 1024u) { const uint32_t slot = __builtin_amdgcn_s_getreg(0x1804) & 3u; for (uint32_t i = 0; i < slot * 2u; ++i) __builtin_amdgcn_s_sleep(127); }\n    prefetch_tile<K, NT>(r, ld, t, toff);\n    if (a.frozen_old_local) clear_stale"),
    "fwd_no_barriers": lambda t: once(t, "      if (!(w0 & kRoundNoBarrier)) __syncthreads();  // else the next round's waves read only their own writes\n      pc += kRoundWords;\n    } else if (opc == OP_GATE2) {",
                                      "      pc += kRoundWords;\n    } else if (opc == OP_GATE2) {"),
    "fwd_no_round_trips": lambda t: once(once(t, "      round_load<R>(tile, T, DB, amp);\n      for (uint32_t i = 0; i < n_inst; ++i) {", "      if (pc == 0) round_load<R>(tile, T, DB, amp);\n      for (uint32_t i = 0; i < n_inst; ++i) {"),
                                         "      round_store<R>(tile, T, DB, amp);\n      if (!(w0 & kRoundNoBarrier)) __syncthreads();", "      if (lane == 77) round_store<R>(tile, T, DB, amp);\n      if (!(w0 & kRoundNoBarrier)) __syncthreads();"),
})

_ABL_HELPERS = '\n// ---- ablation helpers (scripts/ablate/build.py): 16-byte LDS accesses on slot pairs, MFMA filler ----\ntemplate <int R, int... I>\n__device__ __forceinline__ void round_load_b128_(const char* __restrict__ base, uint32_t addr, const uint32_t (&DB)[R],\n                                                 v2f (&a)[1 << R], std::integer_sequence<int, I...>) {\n  ((addr ^= (I ? DB[1 + (I ? __builtin_ctz(I) : 0)] : 0u),\n    [&] { const float4 v = *reinterpret_cast<const float4*>(base + (addr & ~8u));\n          a[2 * (I ^ (I >> 1))] = v2f{v.x, v.y}; a[2 * (I ^ (I >> 1)) + 1] = v2f{v.z, v.w}; }()), ...);\n}\ntemplate <int R, int... I>\n__device__ __forceinline__ void round_store_b128_(char* __restrict__ base, uint32_t addr, const uint32_t (&DB)[R],\n                                                  const v2f (&a)[1 << R], std::integer_sequence<int, I...>) {\n  ((addr ^= (I ? DB[1 + (I ? __builtin_ctz(I) : 0)] : 0u),\n    *reinterpret_cast<float4*>(base + (addr & ~8u)) =\n        make_float4(a[2 * (I ^ (I >> 1))].x, a[2 * (I ^ (I >> 1))].y, a[2 * (I ^ (I >> 1)) + 1].x, a[2 * (I ^ (I >> 1)) + 1].y)), ...);\n}\ntypedef float v4f_abl __attribute__((ext_vector_type(4)));\n__device__ __forceinline__ void mfma_filler(v4f_abl& acc, float x, int count) {\n  for (int i = 0; i < count; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, acc, 0, 0, 0);\n}\n\n}  // namespace\n\n// ================================================================================\n// Forward pass kernel\n'
_ABL_ANCHOR = '}  // namespace\n\n// ================================================================================\n// Forward pass kernel\n'
_FWD_LOAD = '      round_load<R>(tile, T, DB, amp);\n      for (uint32_t i = 0; i < n_inst; ++i) {'
_FWD_STORE = '      round_store<R>(tile, T, DB, amp);\n      if (!(w0 & kRoundNoBarrier)) __syncthreads();'


def _with_helpers(t):
  return once(t, _ABL_ANCHOR, _ABL_HELPERS)


VARIANTS.update({
    # halve the DS instructions of the forward exchange with 16-byte accesses on slot
    # pairs (addresses forced even: wrong amplitudes, the access pattern and instruction count of a layout
    # that keeps a register bit at LDS address bit 0)
    "fwd_b128_store": lambda t: once(_with_helpers(t), _FWD_STORE,
        "      round_store_b128_<R>(reinterpret_cast<char*>(tile), T, DB, amp, iseq<8>{});\n      if (!(w0 & kRoundNoBarrier)) __syncthreads();"),
    "fwd_b128_load": lambda t: once(_with_helpers(t), _FWD_LOAD,
        "      round_load_b128_<R>(reinterpret_cast<const char*>(tile), T, DB, amp, iseq<8>{});\n      for (uint32_t i = 0; i < n_inst; ++i) {"),
    "fwd_b128_both": lambda t: once(once(_with_helpers(t), _FWD_STORE,
        "      round_store_b128_<R>(reinterpret_cast<char*>(tile), T, DB, amp, iseq<8>{});\n      if (!(w0 & kRoundNoBarrier)) __syncthreads();"), _FWD_LOAD,
        "      round_load_b128_<R>(reinterpret_cast<const char*>(tile), T, DB, amp, iseq<8>{});\n      for (uint32_t i = 0; i < n_inst; ++i) {"),
    # the matrix pipe BESIDE the VALU.  A dense 16 x 16 complex block per round is 32
    # v_mfma_f32_32x32x2_f32 = 2048 MFMA-pipe cycles per wave (scripts/micro/mfma_block.hip); the same pipe
    # time as 64 v_mfma_f32_16x16x4_f32 on ONE 4-register accumulator (the kernel keeps its four waves per
    # SIMD), issued after the instances of every round.  fwd_mfma: on top of the unchanged VALU work (what
    # co-issue costs); fwd_no_x_mfma: with the X**t shears compiled out (what the round would cost if the X
    # layer moved to the matrix pipe for free -- no operand staging, no lane swaps).
    "fwd_mfma": lambda t: once(_with_helpers(t), _FWD_STORE,
        "      { v4f_abl macc = {0.f, 0.f, 0.f, 0.f}; mfma_filler(macc, amp[0].x, 64); if (macc.x == 12345.f) amp[0].x += macc.y; }\n" + _FWD_STORE),
    "fwd_no_x_mfma": lambda t: VARIANTS["fwd_mfma"](VARIANTS["fwd_no_x"](t)),
    "fwd_mfma16": lambda t: once(_with_helpers(t), _FWD_STORE,
        "      { v4f_abl macc = {0.f, 0.f, 0.f, 0.f}; mfma_filler(macc, amp[0].x, 16); if (macc.x == 12345.f) amp[0].x += macc.y; }\n" + _FWD_STORE),
})


VARIANTS.update({
    # what the tile I/O of the PRUNED adjoint passes costs (those whose tiles skip finished bits: whole
    # 128-byte lines are moved for a half / quarter / ... of their amplitudes): their loads and stores
    # compiled out -- the time a compacted state layout could approach
    "adj_no_io_pruned": lambda t: once(once(t,
        "  prefetch_tile<K, NT, ROWS>(rp, sp, t, toff);\n  prefetch_tile<K, NT, ROWS>(rl, sl, t, toff);\n",
        "  if (!a.zero_mask) {\n  prefetch_tile<K, NT, ROWS>(rp, sp, t, toff);\n  prefetch_tile<K, NT, ROWS>(rl, sl, t, toff);\n  } else { rp = TileRegs{}; rl = TileRegs{}; rp.p0.x = 1e-3f; rl.p0.y = 1e-3f; }\n"),
        "  } else if (a.flags & PASS_STORE) {\n    const ThreadOff o = thread_offsets<ROWS>(t, tid);",
        "  } else if ((a.flags & PASS_STORE) && !a.zero_mask) {\n    const ThreadOff o = thread_offsets<ROWS>(t, tid);"),
})


VARIANTS.update({
    # the adjoint WITHOUT its HBM traffic (every pass: tiles neither loaded nor stored; constants instead): the compute
    # and exchange time of a pass alone -- with `no_instances` (the I/O and the exchange alone) it says how much of a pass
    # is the sum of the two and how much their maximum (round 4: pass 0 of config 3)
    "adj_no_io": lambda t: once(once(once(t,
        "  prefetch_tile<K, NT, ROWS>(rp, sp, t, toff);\n  prefetch_tile<K, NT, ROWS>(rl, sl, t, toff);\n",
        "  rp = TileRegs{}; rl = TileRegs{}; rp.p0.x = 1e-3f; rl.p0.y = 1e-3f;\n"),
        "  if (a.flags & PASS_RELABEL) {\n    const RelabelCtx rc = relabel_lookup<K>(a, tables, in_local, tid, lane);  // (in flight under the exchange below)",
        "  if ((a.flags & PASS_RELABEL) && tid == 100000) {\n    const RelabelCtx rc = relabel_lookup<K>(a, tables, in_local, tid, lane);"),
        "  } else if (a.flags & PASS_STORE) {\n    const ThreadOff o = thread_offsets<ROWS>(t, tid);",
        "  } else if ((a.flags & PASS_STORE) && tid == 100000) {\n    const ThreadOff o = thread_offsets<ROWS>(t, tid);"),
})


VARIANTS.update({
    # round 5: the paired forward kernel at FIVE waves per SIMD (its LDS -- one 32-KiB exchange tile -- admits five
    # workgroups per CU; 96 registers instead of 111: the compiler spills 16 in the tile prologue)
    # round 5: the reductions with presence tests -- a scalar test + branch per pair / per value and pair (shipped: none)
    "red_tests_pairs": lambda t: once(t, "#define QHBM_RED_TESTS 0\n", "#define QHBM_RED_TESTS 1\n"),
    "red_tests_values": lambda t: once(t, "#define QHBM_RED_TESTS 0\n", "#define QHBM_RED_TESTS 2\n"),
    "fwd2_five_waves": lambda t: once(t,
        "template <int K>\n__global__ __launch_bounds__(1 << (K - 4), adjx_min_waves(K)) void pass_fwd2_kernel(",
        "constexpr int fwd2_min_waves(int K) { return clampi(wg_per_cu(8 << K) * (1 << (K - 4)) / 256, 1, 5); }\n"
        "template <int K>\n__global__ __launch_bounds__(1 << (K - 4), fwd2_min_waves(K)) void pass_fwd2_kernel("),
})


VARIANTS.update({
    # round 5: the adjoint kernel's phases timed with s_memtime (-DQHBM_ADJ_TIMING: the launcher prints cycles per wave and
    # phase after every launch; results stay correct)
    "adj_timing": lambda t: once(t, "#include <hip/hip_runtime.h>\n", "#define QHBM_ADJ_TIMING 1\n#include <hip/hip_runtime.h>\n"),
})


VARIANTS.update({
    # round 5: the PAIRED forward kernel (pass_fwd2_kernel) without its instances / without its tile traffic / without the
    # exchange between rounds: what the skeleton, the I/O and the LDS round trips cost the forward sweep
    "fwd2_no_instances": lambda t: once(t, "      instance_fwd_pair<R, 1>(cur, recs, rec_off, p, q, TL | tile_hi);",
                                        "      if (lane == 77) instance_fwd_pair<R, 1>(cur, recs, rec_off, p, q, TL | tile_hi);"),
    "fwd2_no_io": lambda t: once(once(t, "  prefetch_tile<K, NT>(ra, st_a, t, toff);\n  prefetch_tile<K, NT>(rb, st_b, t, toff);\n",
                                      "  ra = TileRegs{}; rb = TileRegs{}; ra.p0.x = 1e-3f; rb.p0.y = 1e-3f;\n"),
                                 "  if (a.flags & PASS_STORE) {\n    round_store0<R>(T, DB, p);\n    __syncthreads();\n    const ThreadOff o = thread_offsets(t, tid);\n    store_tile<K, NT>(xt, st_a, t, o, tid);",
                                 "  if ((a.flags & PASS_STORE) && tid == 100000) {\n    round_store0<R>(T, DB, p);\n    __syncthreads();\n    const ThreadOff o = thread_offsets(t, tid);\n    store_tile<K, NT>(xt, st_a, t, o, tid);"),
    "fwd2_no_exchange": lambda t: once(t, """    round_store0<R>(T, DB, p);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, p);
    if (sync) __syncthreads();
    round_store0<R>(T, DB, q);
    if (sync) __syncthreads();
    round_load0<R>(Tn, DBn, q);
""", ""),
})


_OBS_LOOP = "        if (live) obs_consume<A>(gr, cur, own, st, k, k0, tb, acc);  // else: the group vanishes on the whole block"
VARIANTS.update({
    # lambda = O psi (wrong lambda): without the masks that leave the block (no gathers: staging + the masks served
    # from LDS + the block's own read and write), without the masks inside the block, without either
    "obs_no_far": lambda t: once(t, _OBS_LOOP, "        if (live && gr.x < 256u * A) obs_consume<A>(gr, cur, own, st, k, k0, tb, acc);").replace(
        "      if (gr.x >= 256u * A && !(gr.same_x & 1u)) obs_gather<A>(cur, ps, j0, gr.x, live);", "", 1),
    "obs_no_near": lambda t: once(t, _OBS_LOOP, "        if (live && gr.x >= 256u * A) obs_consume<A>(gr, cur, own, st, k, k0, tb, acc);"),
    "obs_no_groups": lambda t: once(t, _OBS_LOOP, "        if (live && gr.x == 0x7fffffffu) obs_consume<A>(gr, cur, own, st, k, k0, tb, acc);").replace(
        "      if (gr.x >= 256u * A && !(gr.same_x & 1u)) obs_gather<A>(cur, ps, j0, gr.x, live);", "", 1),
})


def check_variants(names=None):
  """Applies every variant's edit to kernels.hip WITHOUT compiling: {name: error message} of the variants
  whose anchors no longer match the kernel source (tests/test_scripts_cpu.py keeps this empty)."""
  src = kernel_source()
  stale = {}
  for name in names or list(VARIANTS):
    try:
      assert VARIANTS[name](src) != src or name == "base", "the edit changed nothing"
    except (AssertionError, ValueError) as exc:
      stale[name] = f"anchor not found: {str(exc)[:120]!r}"
  return stale


def main(which):
  src = kernel_source()
  failed = {}
  for name in which:
    try:
      text = VARIANTS[name](src)
    except (AssertionError, ValueError) as exc:   # a stale anchor must not stop the other variants
      failed[name] = str(exc)[:200]
      print(f"SKIPPED {name}: its anchor no longer matches kernels.hip: {failed[name]!r}", flush=True)
      continue
    path = os.path.join(CSRC, f"_ablate_{name}.hip")
    with open(path, "w") as f:
      f.write(text)
    obj = os.path.join(OUT, f"{name}.o")
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-disable-promote-alloca-to-vector=1",
             "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
    try:
      subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", path, "-o", obj], cwd=CSRC)
      subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj,
                             os.path.join(CSRC, "engine.o"), os.path.join(CSRC, "schedule.o"),
                             os.path.join(CSRC, "observable.o"), "-o",
                             os.path.join(OUT, f"lib_{name}.so")], cwd=CSRC)
      print("built", name, flush=True)
    except subprocess.CalledProcessError as exc:
      failed[name] = f"compile failed: {exc}"
      print(f"FAILED {name}: {exc}", flush=True)
    finally:
      for tmp in (path, obj):
        if os.path.exists(tmp):
          os.remove(tmp)
  return failed


if __name__ == "__main__":
  sys.exit(1 if main(sys.argv[1:] or list(VARIANTS)) else 0)
