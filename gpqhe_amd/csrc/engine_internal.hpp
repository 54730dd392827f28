// engine_internal.hpp -- the context object behind the opaque gpq_ctx handle.
#pragma once
#include <cstdio>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "tables.hpp"

// Optional per-launch timing with HIP events recorded on the launch stream
// (bench.py's roofline leg).  Off by default: no events are created.
struct gpq_prof_rec { int kernel; hipEvent_t a, b; };
enum { GPQ_K_STRIDED_FWD = 0, GPQ_K_STRIDED_INV, GPQ_K_CONTIG_FWD, GPQ_K_CONTIG_INV, GPQ_K_TENSOR_MID,
       GPQ_K_KEYSWITCH_MID, GPQ_K_POINTWISE, GPQ_K_SMALL, GPQ_K_REFERENCE,
       // the MPI <-> RNS bridge (bridge.hip): rns_decompose, the CRT fast paths, the relinearisation front, its one-pass form, the exact /
       // masked kernels (bridge_reconstruct, bridge_roundfix, bridge_addround, bridge_exactdiv), he_rs
       GPQ_K_DECOMPOSE, GPQ_K_RECONSTRUCT, GPQ_K_RELIN_FRONT, GPQ_K_RELIN_TAIL_FUSED, GPQ_K_BRIDGE_EXACT, GPQ_K_RESCALE, GPQ_K_RELIN_TAIL_DIRECT,
       // bridge_stream.hpp: CRT(d2hat) -> rns_decompose in one kernel; the one-product tail with its addend's CRT in the same kernel
       GPQ_K_CRT_DECOMPOSE, GPQ_K_TAIL_STREAM, GPQ_K_COUNT };

// Constant matrix of the matrix-core CRT fast path for one basis and result width WL (bridge_mfma.hpp)
struct gpq_recon_mfma {
  void *d_bfrag = nullptr;
  uint64_t *d_lk = nullptr, *d_kc = nullptr, *d_pm = nullptr;
  unsigned KS = 0;
  size_t lds_bytes = 0;
};

// CRT constants of one prefix of the prime chain (bridge.hip)
struct gpq_bridge_basis {
  unsigned first = 0, dim = 0, pbits = 0;
  int WP = 0;
  uint64_t *d_phat = nullptr, *d_phat_inv = nullptr, *d_pmult = nullptr, *d_phalf = nullptr, *d_inv128 = nullptr;
  std::vector<uint64_t> h_phat_inv;
  std::vector<uint64_t> h_P;          // the product itself, little-endian words
  std::vector<uint64_t> h_phat;       // [dim][WP]
  std::map<int, gpq_recon_mfma> mfma; // by result width WL
  gpq::LimbTab *d_tabs_scaled = nullptr;   // the context's LimbTab array with n^-1 * (P/p_d)^-1 for the limbs of this basis (ScaledInverse, bridge.hip)
};

// Constant matrix of the matrix-core rns_decompose for (first limb, limbs, words) (bridge_mfma.hpp)
struct gpq_decomp_mfma {
  void *d_bfrag = nullptr;
  uint64_t *d_pk = nullptr;
  unsigned NT = 0, KS = 0;
  size_t lds_bytes = 0;
};

// P^-1 mod p_d for the limbs above a P of dimP limbs (exact division in he_relin / he_swk)
struct gpq_relin_tables {
  uint64_t *d_pinv = nullptr;
  // bridge_relin_front_mfma (bridge_mfma.hpp): constant matrix and per-limb tables, built on first use
  bool front_tried = false;
  void *d_bfrag = nullptr;
  uint64_t *d_lk = nullptr, *d_pk = nullptr, *d_tkp = nullptr, *d_kf = nullptr;
  unsigned NT = 0, KS = 0;
  size_t lds_bytes = 0;
  // the same tables for limbs that arrive multiplied by w_j = P^-1 (Pi'/p_j)^-1, and the per-limb table that makes the key switch's
  // inverse pass deliver them so (bridge.hip: get_relin_front, tail_prescale_mode)
  void *d_bfrag_w = nullptr;
  uint64_t *d_pk_w = nullptr, *d_tkp_w = nullptr;
  gpq::LimbTab *d_tabs_w = nullptr;
  // the one-product tail (bridge.hip: get_tail_direct): constant matrix of floor(Pi' 2^104 / p_d) over all dimB limbs, the per-limb table that
  // makes the key switch deliver CRT-weighted limbs (owned by the basis), the weights and their inverses for bridge_limb_scale
  bool direct_tried = false;
  gpq_recon_mfma direct;
  std::map<unsigned, gpq_recon_mfma> direct_padded;   // the same matrix zero-padded to KS k steps (bridge_stream.hpp's instantiations)
  const gpq::LimbTab *d_tabs_direct = nullptr;
  uint64_t *d_scale = nullptr, *d_unscale = nullptr;
};

// Everything a context builds once and only reads afterwards on the device: the transform tables of upload_tables (engine.hip) and the bridge
// constants bridge.hip builds at first use.  ONE object per prime chain: a context's peer lane (gpq_ctx_clone) points at its parent's, so the second
// lane costs no table memory, no table-building time, and "the peer's tables differ" is not a state the library can be in.  Built and read by the
// one host thread that drives the context (single caller); uploads are synchronous hipMemcpy, so a launch on either lane's stream that
// follows a build on the host sees the table.
struct gpq_table_cache {
  std::map<std::pair<unsigned, unsigned>, gpq_bridge_basis> bases;   // by (first limb, count), built on first use
  std::map<std::pair<unsigned, unsigned>, gpq_relin_tables> relins;  // by (dimP, dimB)
  std::map<std::pair<std::pair<unsigned, unsigned>, unsigned>, gpq_decomp_mfma> decomps;  // by ((first limb, limbs), W)
  size_t device_bytes = 0;            // read-only device memory behind this cache and the transform tables (gpq_table_malloc)
};

struct gpq_ctx {
  int device = 0;
  unsigned logn = 0, n = 0, nprimes = 0;
  unsigned chunk = 32; // polynomials per fused launch group (measured: larger groups amortise launch tails; profiles/r01/v9_sweep_chunk.txt)
  unsigned limb_block = 0; // limbs per fused launch group (0 = all limbs of the slab); gpq_set_limb_block
  // host copies in the reference's own representation (struct rns_ctx, src/poly.h:28-41)
  std::vector<uint64_t> p, pinv_mont, pinv_barr, ninv_mont, psi;
  std::vector<uint64_t> zetas, zetas_inv;  // [nprimes][n], Montgomery form, bit-reversed index
  // device tables (standard form)
  uint64_t *d_w = nullptr, *d_winv = nullptr;
  gpq::TwS *d_ws = nullptr, *d_winvs = nullptr;   // split-twiddle pairs for the first nsplit limbs (modarith.hpp)
  unsigned nsplit = 0;                            // leading limbs that run the split-twiddle butterflies (c < GPQ_SPLIT_CMAX)
  unsigned nsplit_tables = 0;                     // ... and for how many limbs the pair tables exist (gpq_set_limb_classes can only lower nsplit)
  bool low9 = false;                              // n = 2^17: strided passes over 512-coefficient rows, 9 low stages (Lane8<9>)
  unsigned nwide = 0, nwide_max = 0;              // leading limbs with c < GPQ_WIDE_CMAX (<= nsplit): forward stages as ct_bfly_wide
  gpq::LimbTab *d_tabs = nullptr;
  std::vector<gpq::LimbTab> h_tabs;               // host copy (bridge.hip derives tables with pre-scaled n^-1 from it)
  // When set, the inverse strided pass of gpq_he_mul_tensor / gpq_keyswitch reads its per-limb constants here: the same tables
  // with n^-1 (and winv[1] n^-1) multiplied by a CRT weight, so that the limbs leave the transform already scaled for the
  // reconstruction that follows (bridge.hip: ScaledInverse).  Internal to gpq_he_mul / gpq_he_swk.
  const gpq::LimbTab *inv_tabs_override = nullptr;
  gpq_table_cache *cache = nullptr;   // the lazily built bridge constants (owned unless tables_of is set)
  const gpq_ctx *tables_of = nullptr; // a peer lane: d_w / d_winv / d_ws / d_winvs / d_tabs and `cache` are the PARENT's, borrowed for the parent's lifetime
  bool bridge_mfma = true;            // matrix-core decompose / CRT (gpq_set_bridge_mfma(ctx, 0): the integer-VALU kernels)
  unsigned char *d_redo = nullptr;    // per-coefficient "redo exactly" flags of the fast CRT path
  size_t redo_cap = 0;
  std::vector<void *> retired;        // outgrown d_redo buffers: a HIP graph captured earlier may still write them, so they live as long as the context
  unsigned *d_wave_any = nullptr;     // one word per wave of the last bridge_stream.hpp launch: did it flag a coefficient (FlagScope)
  bool lazy_decompose = true;         // gpq_he_mul's own rns_decompose output in (0, 3p): the forward transforms take it (gpq_set_lazy_decompose)
  unsigned debug_force_redo = 0;      // tests (gpq_debug_force_redo)
  int nt_mode = -1;                   // slab traffic of the transform kernels with the nt cache policy: -1 by working set (nt_for), 0 never, 1 always
  bool stream_bridge = true;          // gpq_he_mul / gpq_he_swk: bridge_stream.hpp's fused streaming kernels (gpq_set_stream_bridge(ctx, 0): round 3's separate kernels)
  bool exact_crt = false;             // force the exact CRT kernel (tests)
  bool prescale = true;               // gpq_he_mul / gpq_he_swk: inverse passes write limbs pre-multiplied by (P/p_d)^-1 for the CRT kernels (gpq_set_prescale)
  bool prescale_upper = true;         // ... and the limbs above P by w_j for the relinearisation front
  bool tail_direct = true;            // ... or every limb by the weights of the whole basis: the relinearisation tail as ONE product (bridge.hip: get_tail_direct)
  bool fuse_tail = false;             // gpq_set_fused_tail(ctx, 1): the relinearisation tail in one pass per coefficient (bridge_relin_tail_mfma) -- measured 2 % SLOWER
                                      // than the two-kernel form on the whole he_mul (profiles/r03/v3_fused_tail_ab.txt: both are bound by integer VALU work, not by the
                                      // 60 words per coefficient the fusion saves), kept for the parity tests and as the record of the attempt
  // gpq_he_mul / gpq_he_swk over more than one launch group: every other group runs on a second stream through a PEER context (the parent's
  // read-only tables; its own scratch and flag words: nothing mutable is shared), so that the HBM-bound bridge kernels of one group run beside the issue-bound transforms
  // of the other and launch tails fill (gpq_set_overlap; bridge.hip: peer_lane).  The caller's stream orders the whole call as before.
  int overlap = -1;                   // gpq_set_overlap: -1 (default) = two lanes when the peer's workspace is affordable (kPeerAutoWorkspaceBytes), 0 = never, 1 = always
  bool peer_failed = false;           // the peer (its stream, events, flag words) could not be created once: one lane from then on
  size_t peer_ws_declined = 0;        // a peer workspace of this many bytes could not be allocated: shapes that need as much or more run on one lane, smaller ones still get two
  int debug_peer_fail = 0;            // tests (gpq_debug_fail_peer): 1 = the next attempt to create the peer fails as an allocation would, 2 = the next workspace allocation does
  unsigned last_lanes = 1;            // lanes the last multi-group entry point ran on (gpq_last_lanes)
  std::set<unsigned long long> peer_warm;   // call shapes the peer has run outside a stream capture (gpq_lane_key)
  gpq_ctx *peer = nullptr;
  hipStream_t peer_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void *peer_ws = nullptr;
  size_t peer_ws_bytes = 0;
  unsigned *d_zflag = nullptr;        // per-(polynomial, limb) "output contains a residue 0" flags of gpq_ntt (tables.hpp); zero between calls
  size_t zflag_cap = 0;
  // gpq_debug_zero_watch: copies of the flag words of gpq_ntt's LAST launch group as the forward kernels left them (before ref_zero_redo) and
  // as ref_zero_redo left them (after), taken on the launch stream; [0, zwatch_cap) before, [zwatch_cap, 2 zwatch_cap) after
  unsigned *d_zwatch = nullptr;
  size_t zwatch_cap = 0, zwatch_count = 0;
  // profiling
  bool prof_on = false;
  std::vector<gpq_prof_rec> prof;     // launches recorded since the last reset
  std::vector<hipEvent_t> prof_pool;  // recycled events
};

// Makes `device` current for a scope and puts the caller's device back afterwards (context creation must not leave
// the calling thread on another GPU).
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(int device) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(device); }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// roctx range around a stage of the path (tracing).  The marker library is looked up at run time
// (librocprofiler-sdk-roctx / libroctx64, normally brought in by rocprofv3 --marker-trace); without it a range costs one
// pointer test.  Host-side ranges: they bracket the LAUNCHES of a stage, the kernels themselves appear in the kernel trace.
void gpq_range_push(const char *name);
void gpq_range_pop();
struct StageRange {
  explicit StageRange(const char *name) { gpq_range_push(name); }
  ~StageRange() { gpq_range_pop(); }
  StageRange(const StageRange &) = delete;
  StageRange &operator=(const StageRange &) = delete;
};

// Brackets one kernel launch (or a short run of them) with two events on its stream when profiling is on.
struct ProfScope {
  gpq_ctx *c; hipStream_t s; gpq_prof_rec r; bool on;
  ProfScope(const gpq_ctx *cc, int kernel, hipStream_t st) : c(const_cast<gpq_ctx *>(cc)), s(st), on(cc->prof_on) {
    if (!on) return;
    auto get = [&]() { hipEvent_t e; if (!c->prof_pool.empty()) { e = c->prof_pool.back(); c->prof_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    r.kernel = kernel; r.a = get(); r.b = get();
    (void)hipEventRecord(r.a, s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(r.b, s);
    c->prof.push_back(r);
  }
  ProfScope(const ProfScope &) = delete;
  ProfScope &operator=(const ProfScope &) = delete;
};

int gpq_fail(int code, const char *fmt, ...);
// A second context over the same primes for the peer lane: every read-only device table (twiddles, split pairs, LimbTab, the bridge's constant
// cache) is the parent's, borrowed; its own are only the mutable words (zero flags, redo flags, wave words, scratch).  Default settings, no peer of
// its own; must be destroyed before the parent (gpq_ctx_destroy does).
int gpq_ctx_clone(const gpq_ctx *c, gpq_ctx **out);
// hipMalloc of a read-only table, accounted in the context's table cache (gpq_debug_table_bytes)
inline hipError_t gpq_table_malloc(gpq_ctx *c, void **p, size_t bytes) {
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess && c->cache) c->cache->device_bytes += bytes;
  return e;
}
void gpq_bridge_release(gpq_ctx *c);

// ---------------------------------------------------------------------------
// Two launch groups in flight (gpq_set_overlap).  A call over more than one launch group hands every other group to the context's PEER -- a
// second context over the same primes (the parent's read-only tables by pointer; own scratch and flag words: no mutable state is shared) on a second stream -- by calling the same
// entry point on it for that group's slice of the batch.  Launch tails of one lane fill with the other's work, and the bridge kernels of one
// group (HBM-bound) meet transforms of the other (issue-bound): +3.5 ... +4 % whole he_mul at the headline shape, +12 % at the reference's
// default shape (profiles/r04/v13_two_lanes_ab.txt).  The caller's stream still orders the call as a whole: the peer's stream waits for an
// event recorded on it at entry, and it waits for the peer's last launch before the call returns.  Not taken while profiling (the breakdown
// is that of one stream), for a single group, or when a stream capture would have to allocate.
// ---------------------------------------------------------------------------
struct PeerLane {
  gpq_ctx *c = nullptr;
  hipStream_t s = nullptr;
  void *ws = nullptr;
  // set once the peer stream has been forked off the caller's: the lane is joined on EVERY exit path of the call (an early `return rc` inside the
  // group loop must not leave queued peer work unordered against the caller's stream, nor a stream capture with an unjoined fork)
  gpq_ctx *owner = nullptr;
  hipStream_t caller = nullptr;
  PeerLane() = default;
  PeerLane(const PeerLane &) = delete;
  PeerLane &operator=(const PeerLane &) = delete;
  inline int join();
  ~PeerLane() { (void)join(); }
};
inline void gpq_mirror_settings(gpq_ctx *q, const gpq_ctx *c) {   // whatever decides which kernels a call runs: the peer follows the context it serves
  q->chunk = c->chunk; q->limb_block = c->limb_block; q->nsplit = c->nsplit; q->nwide = c->nwide;
  q->bridge_mfma = c->bridge_mfma; q->lazy_decompose = c->lazy_decompose; q->debug_force_redo = c->debug_force_redo; q->nt_mode = c->nt_mode;
  q->stream_bridge = c->stream_bridge; q->exact_crt = c->exact_crt; q->prescale = c->prescale; q->prescale_upper = c->prescale_upper;
  q->tail_direct = c->tail_direct; q->fuse_tail = c->fuse_tail;
}
// What gpq_set_overlap(ctx, -1) -- the default -- allows the peer's per-group workspace to cost.  Two lanes pay through launch tails, so the gain
// shrinks as the kernels of a group grow: +10-15 % at the reference's default shape (logn 14: 0.1 GB per group) on every device measured,
// +0.15 ... +3.9 % by device at the headline shape (5.6 GB per group), +1.6 % for he_swk at n = 2^17 (7 GB); never a loss (profiles/r04/v13_two_lanes_ab.txt,
// BENCH_r04.json, profiles/r05).  The cost is device memory: this workspace (the tables are shared with the parent since round 6).
constexpr size_t kPeerAutoWorkspaceBytes = (size_t)16 << 30;
// lane->c stays null when the call runs on the caller's stream alone.  `key` names the call shape (entry point, dimensions, group size): a lane is
// only taken inside a stream capture when the peer has already run that very shape outside one (first-use table builds allocate and copy).
// `bytes(ctx)` = the workspace one launch group needs (evaluated on the context itself: the peer shares its tables and follows its settings).  A
// peer that cannot be created is not an error of the call: the HIP error is cleared, the context stops trying, and the call runs on one lane; a
// WORKSPACE that cannot be allocated declines shapes of that size only.
template <typename Bytes>
int gpq_peer_lane(gpq_ctx *c, hipStream_t s, unsigned long long key, Bytes bytes, PeerLane *lane) {
  if (c->overlap == 0 || c->peer_failed || c->prof_on || c->inv_tabs_override) return GPQ_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return GPQ_OK; }
  const bool capturing = cap != hipStreamCaptureStatusNone;
  if (capturing && (!c->peer || !c->peer_warm.count(key))) return GPQ_OK;     // (before bytes(): sizing a new shape may build tables, which allocates)
  // The price first, on the context itself (the peer runs the same settings over the same tables): a shape the auto mode declines, or one whose
  // workspace already failed to allocate, never creates the peer and costs nothing.
  const size_t need = bytes(c);
  if (!need) return GPQ_OK;
  if (c->overlap < 0 && need > kPeerAutoWorkspaceBytes) return GPQ_OK;        // auto: not at this price
  if (c->peer_ws_declined && need >= c->peer_ws_declined) return GPQ_OK;      // this size did not fit once
  auto give_up = [&](const char *what) {                                       // one lane from now on, for this context
    (void)hipGetLastError();
    c->peer_failed = true;
    fprintf(stderr, "gpqhe_hip: second lane unavailable (%s): this context continues on one lane\n", what);
    return (int)GPQ_OK;
  };
  if (!c->peer) {
    if (c->debug_peer_fail == 1) return give_up("gpq_debug_fail_peer");
    if (!c->peer_stream && hipStreamCreateWithFlags(&c->peer_stream, hipStreamNonBlocking) != hipSuccess) return give_up("hipStreamCreateWithFlags");
    if (!c->ev_fork && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) return give_up("hipEventCreateWithFlags");
    if (!c->ev_join && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) return give_up("hipEventCreateWithFlags");
    if (gpq_ctx_clone(c, &c->peer) != GPQ_OK) { c->peer = nullptr; return give_up("peer context"); }   // last: a context with a peer has its stream and events
  }
  gpq_mirror_settings(c->peer, c);
  if (need > c->peer_ws_bytes) {
    if (capturing) return GPQ_OK;
    void *fresh = nullptr;
    if (c->debug_peer_fail == 2 || hipMalloc(&fresh, need) != hipSuccess) {   // this shape only: smaller ones keep their second lane, nothing else is given up
      (void)hipGetLastError();
      c->peer_ws_declined = need;
      fprintf(stderr, "gpqhe_hip: no room for a second lane's workspace of %zu bytes: shapes of this size run on one lane\n", need);
      return GPQ_OK;
    }
    if (c->peer_ws) c->retired.push_back(c->peer_ws);             // a graph captured earlier may still use it
    c->peer_ws = fresh; c->peer_ws_bytes = need;
  }
  hipError_t e = hipEventRecord(c->ev_fork, s);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->peer_stream, c->ev_fork, 0);
  if (e != hipSuccess) return gpq_fail(GPQ_ERR_HIP, "forking the peer stream: %s", hipGetErrorString(e));
  if (!capturing) c->peer_warm.insert(key);
  lane->c = c->peer; lane->s = c->peer_stream; lane->ws = c->peer_ws; lane->owner = c; lane->caller = s;
  return GPQ_OK;
}
inline int PeerLane::join() {
  if (!c || !owner) return GPQ_OK;
  hipError_t e = hipEventRecord(owner->ev_join, s);
  if (e == hipSuccess) e = hipStreamWaitEvent(caller, owner->ev_join, 0);
  c = nullptr; owner = nullptr;
  return e == hipSuccess ? (int)GPQ_OK : gpq_fail(GPQ_ERR_HIP, "joining the peer stream: %s", hipGetErrorString(e));
}
inline int gpq_peer_join(gpq_ctx *, hipStream_t, PeerLane &lane) { return lane.join(); }
inline unsigned long long gpq_lane_key(unsigned entry, unsigned a, unsigned b, unsigned c3, unsigned d, unsigned e) {
  unsigned long long h = 0xcbf29ce484222325ull;
  for (unsigned v : {entry, a, b, c3, d, e}) { h ^= v; h *= 0x100000001b3ull; }
  return h;
}

