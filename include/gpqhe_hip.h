/*
 * gpqhe_hip.h -- C ABI of libgpqhe_hip.so, the MI355X engine behind GPQHE's
 * RNS/NTT hot path.  Plain C: pointers, sizes, opaque handles; no HIP or
 * torch types (a hipStream_t travels as void*).
 *
 * Two groups of entry points:
 *
 *  (1) Reference-named drop-in symbols (gpqhe_hip_compat.h): `ntt`, `invntt`,
 *      `poly_rns_mul`, `poly_rns_add`, `montgomery_*`, `barrett_*` with the
 *      reference's exact signatures (host pointers, one limb, synchronous),
 *      plus the north-star aliases `poly_ntt`, `poly_invntt`.
 *
 *  (2) The slab API below: whole limb-major slabs `uint64_t[batch][dim][n]`
 *      resident in HBM, asynchronous on a caller-supplied stream.  These are
 *      what the reference's limb loops (src/poly.c:96-103, src/he-mult.c:
 *      116-138 and :58-66, src/he-automorphism.c:59-67) become once the loop
 *      body is a kernel launch instead of a per-limb function call.
 *
 * Value contract (same as the reference): inputs are
 * canonical residues < p_d, outputs canonical; forward output order is the
 * reference's bit-reversed order; slab element (d, i) lives at d*n + i
 * (src/poly.c:99, src/rns.c:67).
 *
 * Concurrency: like the reference (SURVEY.md 8b "Threading: none") a context serves one caller at
 * a time: calls on one context must be issued to one stream at a time (the context owns small
 * scratch that successive launches reuse in stream order).  Use one context per stream otherwise.
 *
 * Errors: the slab API returns GPQ_OK or a negative code and records a
 * message (gpq_last_error); it never aborts.  The drop-in symbols keep the
 * reference's convention: void return, errno=EINVAL + message + abort() on
 * misuse (src/reduce.c:95-100).
 */
#ifndef GPQHE_HIP_H
#define GPQHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPQ_OK 0
#define GPQ_ERR_INVALID (-1)     /* bad argument / unsupported ring size        */
#define GPQ_ERR_HIP (-2)         /* HIP runtime error, see gpq_last_error()      */
#define GPQ_ERR_UNSUPPORTED (-3) /* prime outside the 2^59+c family the kernels fold */
#define GPQ_ERR_NOMEM (-4)

typedef struct gpq_ctx gpq_ctx;

/* ---- context: replaces polyctx_init's RNS part ------------------------------
 * Builds the reference's prime chain and per-prime tables for ring degree
 * n = 2^logn and uploads them to `device`:
 *   primes      p_0 < p_1 < ..., next prime == 1 (mod 2n) above 2^59+1   src/precomp.c:358, :372-376
 *   constants   pinv_mont, pinv_barr, ninv                               src/precomp.c:246-248
 *   twiddles    psi = g^((p-1)/2n), g least primitive root; tables in
 *               bit-reversed order                                       src/precomp.c:206-242, :251-263
 * `nprimes` plays polyctx.dimub (src/precomp.c:357).  logn in [1,17].
 */
int gpq_ctx_create(gpq_ctx **out, unsigned logn, unsigned nprimes, int device);

/* Same, but adopting tables the caller already has in the reference's format
 * (struct rns_ctx fields p / zetas / zetas_inv, Montgomery form): what the
 * drop-in symbols use so that results follow the caller's tables exactly. */
int gpq_ctx_create_from_tables(gpq_ctx **out, unsigned logn, unsigned nprimes, const uint64_t *primes,
                               const uint64_t *const *zetas_mont, const uint64_t *const *zetas_inv_mont, int device);
void gpq_ctx_destroy(gpq_ctx *ctx);

unsigned gpq_ctx_logn(const gpq_ctx *ctx);
unsigned gpq_ctx_nprimes(const gpq_ctx *ctx);
int gpq_ctx_device(const gpq_ctx *ctx);
/* Per-prime constants exactly as the reference's struct rns_ctx holds them
 * (src/poly.h:28-41): which = 0 p, 1 pinv_mont, 2 pinv_barr, 3 ninv, 4 psi. */
uint64_t gpq_ctx_const(const gpq_ctx *ctx, unsigned d, int which);
/* Host copies of rns->zetas / rns->zetas_inv (Montgomery form), n entries. */
const uint64_t *gpq_ctx_zetas(const gpq_ctx *ctx, unsigned d, int inverse);

/* polyctx.dimub for a modulus of logq bits, src/precomp.c:357 (logqub=logq). */
unsigned gpq_dimub(unsigned logn, unsigned logq);

const char *gpq_last_error(void);

/* ---- device memory helpers (for hosts without another allocator) ---------- */
int gpq_malloc(void **dptr, size_t bytes);
int gpq_free(void *dptr);
int gpq_upload(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int gpq_download(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int gpq_copy(void *dst_dev, const void *src_dev, size_t bytes, void *stream);      /* device to device, same device */
int gpq_stream_sync(void *stream);
/* Stream-to-stream ordering without blocking the host: work queued on `waiter` AFTER this call starts only when everything queued on `on` BEFORE it
 * has finished.  With it a plain-C host pipelines sub-batches over three streams -- gpq_upload of k+1 | the stages of k | gpq_download of k-1 -- and the
 * full-duplex link carries uploads and downloads at once (PCIe-inclusive he_mul/s 302 -> 470 on one MI355X; tests/c/shard_host.c `pipe`). */
int gpq_stream_wait(void *waiter, void *on);
/* Several devices from one C program: gpq_malloc and gpq_stream_create act on the calling thread's current device
 * (gpq_set_device); a context belongs to the device given to gpq_ctx_create and is used with streams and buffers of that
 * device.  Independent ciphertexts shard over devices without any exchange (tests/c/shard_host.c). */
int gpq_device_count(void);
/* Yardstick for the HBM-bound kernels (bench.py `copy_rate`): a plain stream over `bytes` (multiple of 16) with the library's own access shape --
 * 16 bytes per lane, `blocks` persistent workgroups of `threads` threads, `unroll` (1, 2, 4, 8) independent accesses in flight per lane.
 * kind 0: dst <- src, 1: read src only, 2: write dst only.  Device pointers; asynchronous on `stream`. */
int gpq_probe_stream(void *dst, const void *src, size_t bytes, int kind, unsigned blocks, unsigned threads, unsigned unroll, void *stream);
int gpq_set_device(int device);
int gpq_stream_create(void **stream);        /* non-blocking stream on the current device */
int gpq_stream_destroy(void *stream);
/* Page-locked host memory: gpq_upload / gpq_download from / to it are asynchronous DMA; with pageable memory the copies
 * are staged and a download waits for the work queued before it. */
int gpq_malloc_host(void **hptr, size_t bytes);
int gpq_free_host(void *hptr);
/* Host placement on a multi-socket node (no counterpart in the reference, which is single-threaded: SURVEY.md 8e).  A C host that drives several
 * GPUs runs one worker thread per device and calls gpq_bind_thread_to_device(device) as the thread's FIRST action -- before gpq_set_device -- so
 * that the thread, and the page-locked buffers it allocates next, sit on the socket the GPU hangs off.  Pure sysfs, no HIP call: HIP device i is
 * the i-th GPU node of /sys/class/kfd/kfd/topology/nodes whose render node this process can open, after ROCR_VISIBLE_DEVICES and HIP_VISIBLE_DEVICES
 * (integer lists); its CPUs are /sys/class/drm/renderD<minor>/device/local_cpulist.  Returns the number of CPUs the thread is now confined to, or 0
 * when nothing was changed (unknown topology, one memory domain, already confined, not allowed): never an error.
 * gpq_device_local_cpus only looks: the count and, in `cpulist`, the kernel's list text ("0-47,96-143"); `sysfs_root` NULL = "/" (tests hand it a
 * fake tree).  The rank processes of bench.py do the same in Python (gpqhe_amd/affinity.py) before their first HIP call. */
int gpq_bind_thread_to_device(int device);
int gpq_device_local_cpus(int device, const char *sysfs_root, char *cpulist, size_t cap);

/* ---- slab operations ------------------------------------------------------
 * All slabs are device pointers to uint64_t[batch][dim][n] using primes
 * 0..dim-1 of the context.  `stream` is a hipStream_t (NULL = default).
 * Kernels launch on the calling thread's CURRENT device, which must be the context's (gpq_set_device(gpq_ctx_device(ctx)));
 * a call from a thread on another device returns GPQ_ERR_INVALID instead of launching there.  A context and the calls on it
 * are not thread-safe: one host thread per context at a time -- and ONE STREAM per context at a time: a context carries scratch
 * that its launches share in stream order (gpq_ntt's zero flags, the per-coefficient redo flags and per-wave words of the bridge,
 * the table override of gpq_he_mul's inverse passes).  Work that should overlap on several streams uses one context per stream
 * (contexts of the same ring share nothing mutable; tools/ntt_stream_probe.py, tests/c/shard_host.c do exactly that).
 */

/* ntt / invntt over every limb of every polynomial, in place.
 * Replaces the per-limb calls `ntt(a, rns)` / `invntt(a, rns)`, src/ntt.c:37,54.
 * Inputs are canonical residues in [0, p_d) (what rns_decompose and poly_rns_mul produce); gpq_invntt also takes the word p_d for a
 * residue 0 -- the domain of the reference's invntt, and what gpq_ntt hands out -- so gpq_ntt's output goes straight into gpq_invntt
 * (tests/test_ntt_zero_repr_gpu.py: round trips and limbs that are p_d throughout, every kernel class).  Outputs are the reference's
 * words: gpq_invntt's are canonical; gpq_ntt's are canonical except that a residue 0 is stored as p_d wherever
 * src/ntt.c:47 stores it so (a sum leg x + t == p is kept as p, and survives while its partner's product is 0):
 * limbs whose output contains a zero are redone with the reference's own arithmetic (ref_zero_redo, ntt_kernels.hpp). */
int gpq_ntt(gpq_ctx *ctx, uint64_t *slab, unsigned dim, unsigned batch, void *stream);
int gpq_invntt(gpq_ctx *ctx, uint64_t *slab, unsigned dim, unsigned batch, void *stream);
/* src/ntt.c:37-52 (inverse = 0) or :54-73 executed as written on every limb, for ANY 64-bit input words (the reference's
 * unsigned wrap-around included): the slow kernel gpq_ntt redoes flagged limbs with, what the drop-in `ntt` / `invntt`
 * take for inputs outside [0, p), and an in-device cross-check of the two-pass kernels. */
int gpq_ntt_reference(gpq_ctx *ctx, uint64_t *slab, unsigned dim, unsigned batch, int inverse, void *stream);

/* r = a (*) b and r = a + b, coefficient-wise mod p_d; r may alias a or b.
 * Replace poly_rns_mul / poly_rns_add, src/poly.c:71-82 (decl src/poly.h:84-85). */
int gpq_rns_mul(gpq_ctx *ctx, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned dim, unsigned batch, void *stream);
int gpq_rns_add(gpq_ctx *ctx, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned dim, unsigned batch, void *stream);

/* Limb loop of poly_mul, src/poly.c:96-103 (without rns_decompose):
 * r = invntt(ntt(a) (*) ntt(b)).  a and b are distinct slabs and are overwritten (scratch: for n >= 2^13 the loop runs
 * as three kernels -- strided pass, fused low stages + product, strided pass -- and leaves them half transformed);
 * r may be a or b. */
int gpq_poly_mul_rns(gpq_ctx *ctx, uint64_t *r, uint64_t *a, uint64_t *b, unsigned dim, unsigned batch, void *stream);
/* Limb loop of he_mulpt, src/he-mult.c:179-185: r0 = invntt(ntt(m) (*) ntt(x0)), r1 = invntt(ntt(m) (*) ntt(x1)).
 * m, x0, x1 are distinct slabs and are overwritten; r0 / r1 may be x0 / x1. */
int gpq_mulpt_rns(gpq_ctx *ctx, uint64_t *r0, uint64_t *r1, uint64_t *m, uint64_t *x0, uint64_t *x1, unsigned dim, unsigned batch,
                  void *stream);

/* The fused operations process the batch in groups of `chunk` polynomials so
 * that the scratch stays bounded (default 32; larger groups amortise launch tails). */
int gpq_set_chunk(gpq_ctx *ctx, unsigned chunk);
/* ... and, inside a group of polynomials, in blocks of `limbs` limbs (0 = all limbs at once, the default): the three
 * kernels of a block run back to back, so a block of chunk x limbs x 7 slabs x n x 8 bytes that fits the 256 MiB
 * Infinity Cache is re-read from there instead of from HBM.  Results do not depend on either setting. */
int gpq_set_limb_block(gpq_ctx *ctx, unsigned limbs);
/* Butterfly classes (modarith.hpp): by default every limb runs the cheapest arithmetic its prime p = 2^59 + c admits --
 * the first limbs (c < 2^27) the wide-split butterflies, the limbs up to c < 2^29/3 the split-twiddle ones, the rest the
 * 7-multiply ones.  This call restricts the first two classes to the first `wide` / `split` limbs (clamped to what the
 * chain admits; (0, 0) = the general butterflies everywhere).  Results are bit-identical for every setting; it exists so
 * that each class can be run on every limb and compared (tests/test_ntt_gpu.py). */
int gpq_set_limb_classes(gpq_ctx *ctx, unsigned wide, unsigned split);

/* Bytes of scratch the two fused operations below need for this shape. */
size_t gpq_tensor_workspace_bytes(const gpq_ctx *ctx, unsigned dim, unsigned batch);
size_t gpq_keyswitch_workspace_bytes(const gpq_ctx *ctx, unsigned dim, unsigned batch);

/* Tensor stage of he_mul, src/he-mult.c:116-138 without the rns_decompose
 * calls: a0,a1,b0,b1 stand for the decomposed ct1.c0, ct1.c1, ct2.c0, ct2.c1;
 *   d0 = a0*b0, d2 = a1*b1, d1 = a0*b1 + a1*b0   (negacyclic, per limb).
 * Inputs are preserved.  Outputs may not alias inputs.  b0 == a0 and b1 == a1 (a squaring: he_mul(&ct, &ct, &ct, rlk),
 * src/he-algo.c:151) is recognised and runs two forward transforms instead of four; the results are the same residues. */
int gpq_he_mul_tensor(gpq_ctx *ctx, uint64_t *d0, uint64_t *d1, uint64_t *d2,
                      const uint64_t *a0, const uint64_t *a1, const uint64_t *b0, const uint64_t *b1,
                      unsigned dim, unsigned batch, void *workspace, void *stream);

/* Key-switch inner product of he_relin / he_swk, src/he-mult.c:58-66 ==
 * src/he-automorphism.c:59-67 without rns_decompose: x is the decomposed d2
 * (or d1); evk0/evk1 are ONE key's NTT-domain slabs uint64_t[dim][n] as built
 * by he_genswk (src/he-kem.c:103-110), shared by the whole batch.
 *   c0 = invntt(ntt(x) (*) evk0), c1 = invntt(ntt(x) (*) evk1).  x preserved. */
int gpq_keyswitch(gpq_ctx *ctx, uint64_t *c0, uint64_t *c1, const uint64_t *x,
                  const uint64_t *evk0, const uint64_t *evk1,
                  unsigned dim, unsigned batch, void *workspace, void *stream);

/* ---- MPI <-> RNS bridge ---------------------------------
 * The reference keeps coefficients as libgcrypt MPIs.  On the device a polynomial of
 * big integers is a "big slab": uint64_t[batch][W][n], word j of coefficient i at
 * j*n + i, little-endian words, two's complement over 64*W bits (centred
 * coefficients are signed).  The entry points with a `logq` / `logql` / `logDelta` argument are the
 * tuned ones for power-of-two moduli, as in every parameter set of the reference's tests
 * (tests/gpqhe.c:1349-1352, tests/polymul.c:84-87); the `*_general` entry points further down take
 * any modulus as little-endian words and any 64-bit Delta.
 */
unsigned gpq_big_words(unsigned bits);
/* rns->phat_invmp[d] of the node with `dim` limbs (src/precomp.c:288-289); bit length of its P. */
uint64_t gpq_ctx_phat_invmp(gpq_ctx *ctx, unsigned dim, unsigned d);
unsigned gpq_ctx_pbits(gpq_ctx *ctx, unsigned dim);

/* rns_decompose for every limb at once, src/rns.c:37-48: slab[k][d][i] = big[k][.][i] mod p_d (>= 0). */
int gpq_rns_decompose(gpq_ctx *ctx, uint64_t *slab, const uint64_t *big, unsigned W, unsigned dim, unsigned batch, void *stream);
/* The same for limbs first .. first+count-1 only (the reference's rns_decompose is per limb): slab[k][d][i], d < count. */
int gpq_rns_decompose_limbs(gpq_ctx *ctx, uint64_t *slab, const uint64_t *big, unsigned W, unsigned first, unsigned count,
                            unsigned batch, void *stream);
/* rns_reconstruct, src/rns.c:60-75, for ONE coefficient: `dim` reduced residues in host memory -> value in [0, P) as
 * Wout host words (64*Wout > bits of P).  Synchronous; the per-coefficient spelling of the reference, kept for its callers. */
int gpq_rns_reconstruct_one(gpq_ctx *ctx, uint64_t *words, unsigned Wout, const uint64_t *residues, unsigned dim);
/* poly_rns2mpi, src/poly.c:109-120, with q = 2^logq: rns_reconstruct (src/rns.c:60-75), centre
 * mod P, centre mod q.  logq = 0 stops after centring mod P (Wout must then hold P's bits + 1). */
int gpq_rns_reconstruct(gpq_ctx *ctx, uint64_t *big, unsigned Wout, const uint64_t *slab, unsigned dim, unsigned batch,
                        unsigned logq, void *stream);
/* gpq_rns_reconstruct takes a low-word fast path when q is a power of two shorter than P and redoes the
 * (rare) coefficients whose rounding it cannot decide with the exact full-width kernel; this forces the
 * exact kernel for everything (used by the tests to cross-check the two). */
int gpq_set_exact_crt(gpq_ctx *ctx, int on);
/* gpq_he_mul / gpq_he_swk: 1 (default) = the bridge as streaming kernels (gpqhe_amd/csrc/bridge_stream.hpp): poly_rns2mpi(d2) -> rns_decompose
 * (src/he-mult.c:140, :59) in one kernel, and the relinearisation tail (:67-77) as one product that makes its addend d0 / d1 (:139, :141) from the
 * limbs on the spot -- d0, d1, d2 never exist as words; 0 = round 3's separate CRT, decompose and tail kernels.  Same words (the tests run both). */
int gpq_set_stream_bridge(gpq_ctx *ctx, int on);
/* gpq_he_mul / gpq_he_swk / gpq_he_mul_tensor / gpq_keyswitch over more than one launch group (batch > gpq_set_chunk's group): with two LANES every
 * other group runs on a second, internal stream through an internal peer context, so that the launch tails of one group fill with the other's
 * kernels; the caller's stream still orders the call as a whole (work queued before it is waited for, work queued after it waits for both lanes).
 * `on` = -1 (default): two lanes when the peer's per-group workspace costs at most 16 GiB; 0: never; 1: always.  Same words either way.
 * COST: a workspace for one launch group (the size gpq_*_workspace_bytes reports for a batch of one group: 5.6 GB for gpq_he_mul at the headline
 * shape), allocated at the first multi-group call that will use it.  The peer borrows every read-only device table of the context (twiddles, split
 * pairs, per-limb constants, bridge constants: gpq_debug_table_bytes) and owns only its flag words and scratch.  GAIN: +10-15 % at the reference's default shape (logn 14) on every device measured; at the headline
 * shape it depends on the device (+0.15 % ... +3.9 %: a group's kernels are long, the lanes share one power cap); never a loss.  If the peer cannot
 * be created the context continues on one lane; if a workspace cannot be allocated, shapes of that size do (one line on stderr, no error).  Not taken while gpq_profile is on.
 * gpq_last_lanes: the lanes (1 or 2) the last such call on this context ran on. */
int gpq_set_overlap(gpq_ctx *ctx, int on);
unsigned gpq_last_lanes(const gpq_ctx *ctx);
/* Tests: 1 = the next attempt to create the peer lane fails as an allocation would (the call runs on one lane, one line on stderr, and the context
 * stops trying); 2 = the next allocation of the peer's WORKSPACE fails (shapes that need that many bytes or more run on one lane from then on,
 * smaller ones keep two); 3 = stop failing but remember what was declined; 0 = back to normal (and the context may try again). */
int gpq_debug_fail_peer(gpq_ctx *ctx, int on);
/* Tests / accounting: read-only device memory (bytes) behind the context -- which = 0: the transform tables and every bridge constant built so far,
 * owned by the context; 1: what its peer lane owns of the same kind (0 by construction: the peer borrows); 2: 1 when a peer lane exists; 3: 1 when
 * every table pointer of the peer IS the context's. */
size_t gpq_debug_table_bytes(const gpq_ctx *ctx, int which);
/* gpq_he_mul: 1 (default) = its internal rns_decompose launches leave residues in (0, 3p), which the forward transforms behind them accept;
 * 0 = canonical residues.  Same results. */
int gpq_set_lazy_decompose(gpq_ctx *ctx, int on);
/* Cache policy of the slab loads / stores of the transform kernels (gpq_ntt, gpq_invntt, gpq_poly_mul_rns, gpq_mulpt_rns, gpq_he_mul_tensor,
 * gpq_keyswitch and everything built on them): -1 (default) = non-temporal when a launch group's slabs exceed the Infinity Cache (288 MiB is the
 * threshold), the default policy when they fit; 0 = never; 1 = always.  Never changes a word. */
int gpq_set_nt_policy(gpq_ctx *ctx, int mode);
/* Tests: the streaming kernels also flag every coefficient whose index is a multiple of `every` for the exact kernels behind them (0 = off). */
int gpq_debug_force_redo(gpq_ctx *ctx, unsigned every);
/* Tests: the zero watch of gpq_ntt (the reference stores p, not 0, at some positions of a forward transform, src/ntt.c:45-48; the forward
 * kernels flag every (polynomial, limb) whose output holds a residue 0 and a kernel that executes src/ntt.c as written redoes those limbs).
 * With the watch on, each launch group of gpq_ntt leaves a copy of its flag words as the forward kernels wrote them and a copy as the redo
 * kernel left them; gpq_debug_zero_flags waits for the device and hands out both for the LAST launch group (word [polynomial * dim + limb];
 * `before` 1 = flagged, `after` all 0), returning the number of words of that group (at most `capacity` are written), -1 without a watch. */
int gpq_debug_zero_watch(gpq_ctx *ctx, int on);
long gpq_debug_zero_flags(gpq_ctx *ctx, unsigned *before, unsigned *after, size_t capacity);
/* With gpq_set_prescale(ctx, 2): the tail of he_relin / he_swk as two kernels with Q's residues in memory between them (0, default) or in one pass per coefficient (1:
 * measured 2 % slower on the whole he_mul -- both forms are bound by integer VALU work); same results. */
int gpq_set_fused_tail(gpq_ctx *ctx, int on);
/* gpq_he_mul / gpq_he_swk: the inverse transforms hand the kernels that follow limbs already multiplied by their CRT weights -- 3 (default): the key
 * switch's limbs carry the weights of its WHOLE basis and the relinearisation tail is one product (quotient, rounding and centring together);
 * 2: (P/p_d)^-1 on the limbs of each basis and w_j = P^-1 (Pi'/p_j)^-1 on the limbs above P for the relinearisation front; 1: the former only;
 * 0: neither.  Same results. */
int gpq_set_prescale(gpq_ctx *ctx, int on);
/* rns_decompose is a product of the coefficients' bytes with a fixed matrix (256^k mod p_j) and runs on the matrix cores
 * (v_mfma_i32_32x32x32_i8, exact) by default; 0 selects the integer-VALU kernel instead (the tests cross-check the two). */
int gpq_set_bridge_mfma(gpq_ctx *ctx, int on);

/* poly_mul, src/poly.c:84-107 (decl src/poly.h:86-87), q = 2^logq, on big slabs of W words. */
size_t gpq_poly_mul_workspace_bytes(const gpq_ctx *ctx, unsigned dim, unsigned batch);
int gpq_poly_mul(gpq_ctx *ctx, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned W, unsigned dim, unsigned logq,
                 unsigned batch, void *workspace, void *stream);
/* The same two for ANY modulus q (little-endian words q_words[0..Lq)): he_genswk calls poly_mul with
 * q = P*q_L (src/he-kem.c:95).  Second centring by multiword Barrett reduction; a key-generation path,
 * not tuned (Horner over Lq-word blocks, one Barrett step each). */
size_t gpq_poly_mul_general_workspace_bytes(gpq_ctx *ctx, unsigned dim, unsigned batch);
int gpq_rns_reconstruct_general(gpq_ctx *ctx, uint64_t *big, unsigned Wout, const uint64_t *slab, unsigned dim, unsigned batch,
                                const uint64_t *q_words, unsigned Lq, void *scratch, void *stream);
int gpq_poly_mul_general(gpq_ctx *ctx, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned W, unsigned dim,
                         const uint64_t *q_words, unsigned Lq, unsigned batch, void *workspace, void *stream);

/* he_rs, src/he-rescale.c:33-54 (decl src/gpqhe.h:136), Delta = 2^logDelta, q_l = 2^logql: c0, c1 of
 * `batch` ciphertexts in place: mpi_rdiv by Delta then mpi_smod q_l.  The l / nu / B bookkeeping
 * (:36-38) stays with the caller.  gpq_he_rescale is the north-star spelling of the same call. */
int gpq_he_rs(gpq_ctx *ctx, uint64_t *c0, uint64_t *c1, unsigned W, unsigned logDelta, unsigned logql, unsigned batch, void *stream);
int gpq_he_rescale(gpq_ctx *ctx, uint64_t *c0, uint64_t *c1, unsigned W, unsigned logDelta, unsigned logql, unsigned batch, void *stream);

/* Limb counts the reference derives from the modulus chain for q_L = 2^logqL, q_l = 2^logql:
 * dimP = hectx.dim (src/precomp.c:401; hectx.P is the product of the first dimP primes),
 * dimA = tensor stage of he_mul (src/he-mult.c:99), dimB = he_relin / he_swk (src/he-mult.c:51,
 * src/he-automorphism.c:52), dimevk (src/precomp.c:407).  Any output pointer may be NULL. */
int gpq_he_dims(gpq_ctx *ctx, unsigned logqL, unsigned logql, unsigned *dimP, unsigned *dimA, unsigned *dimB, unsigned *dimevk);

/* he_mul, src/he-mult.c:88-156 (decl src/gpqhe.h:147), on big slabs of W words with q_l = 2^logql:
 * decompose the four polynomials to dimA limbs, tensor stage, poly_rns2mpi of d0,d1,d2, then he_relin
 * (:40-85): decompose d2 to dimB limbs, key-switch with rlk (NTT-domain slabs of >= dimB limbs,
 * src/he-kem.c:103-110), c = rdiv(poly_rns2mpi(.), P) + d, centred mod q_l.  The l/nu/B bookkeeping
 * (:92-95) stays with the caller.  Outputs may not alias inputs. */
size_t gpq_he_mul_workspace_bytes(gpq_ctx *ctx, unsigned W, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch);
int gpq_he_mul(gpq_ctx *ctx, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *ct1c0, const uint64_t *ct1c1,
               const uint64_t *ct2c0, const uint64_t *ct2c1, const uint64_t *rlk0, const uint64_t *rlk1, unsigned W,
               unsigned logql, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream);
/* he_mul followed by he_rs (src/he-mult.c:88-156, then src/he-rescale.c:33-54 with Delta = 2^logDelta and q_(l-1) = 2^(logql - logDelta)): the same words
 * as gpq_he_mul + gpq_he_rs(out_c0, out_c1, W, logDelta, logql - logDelta), with the rounding division applied by the relinearisation tail on the way
 * out (1 <= logDelta <= 63 and the streaming tail; every other case runs the two calls), so that the outputs are not read and written once more:
 * BASELINE configs[2] is "he_mul + he_rescale".  Workspace as gpq_he_mul. */
int gpq_he_mul_rs(gpq_ctx *ctx, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *ct1c0, const uint64_t *ct1c1,
                  const uint64_t *ct2c0, const uint64_t *ct2c1, const uint64_t *rlk0, const uint64_t *rlk1, unsigned W,
                  unsigned logql, unsigned dimA, unsigned dimB, unsigned dimP, unsigned logDelta, unsigned batch, void *workspace, void *stream);

/* The tail of he_relin / he_swk on its own, src/he-mult.c:67-77: chat = key-switch output slab of dimB
 * limbs, d = big slab added afterwards (NULL: none, as for c1 in he_swk):
 *   out = mpi_smod(mpi_addm(mpi_rdiv(poly_rns2mpi(chat, P*q_l), P), d, q_l), q_l),  q_l = 2^logql. */
size_t gpq_relin_tail_workspace_bytes(gpq_ctx *ctx, unsigned W, unsigned dimB, unsigned dimP, unsigned batch);
int gpq_relin_tail(gpq_ctx *ctx, uint64_t *out, const uint64_t *chat, const uint64_t *d, unsigned W, unsigned logql,
                   unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream);
/* Big slabs between the kernels' layout (word j of coefficient i at j*n + i) and rows of W words per coefficient (i*W + j), the layout a host
 * fills or reads sequentially; to_rows = 0: rows -> words, 1: words -> rows.  n >= 64, dst != src, `batch` polynomials one after another. */
int gpq_big_transpose(gpq_ctx *ctx, uint64_t *dst, const uint64_t *src, unsigned W, unsigned batch, int to_rows, void *stream);
/* out = a + b (mode 0), a - b (mode 1), -a (mode 2; b unused) on `polys` big slabs of W words per coefficient (two's complement,
 * wrapping at 2^(64 W); out may alias a or b): the arithmetic of he_add / he_sub / he_addpt / he_subpt / he_neg before their mpi_smod
 * (src/he-add.c:32-140); gpq_he_rs with logDelta = 0 (or gpq_he_rs_general with delta = 1) is that mpi_smod. */
int gpq_big_addsub(gpq_ctx *ctx, uint64_t *out, const uint64_t *a, const uint64_t *b, unsigned W, unsigned polys, int mode, void *stream);
/* The same with `chat` given up as scratch (overwritten): the tail as ONE matrix-core product over all dimB limbs (what gpq_he_mul / gpq_he_swk use
 * internally, where the key switch already delivers CRT-weighted limbs); same results. */
int gpq_relin_tail_overwriting(gpq_ctx *ctx, uint64_t *out, uint64_t *chat, const uint64_t *d, unsigned W, unsigned logql,
                               unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream);

/* he_swk, src/he-automorphism.c:40-85: key-switch d1 with swk and add d0 into c0, big slabs, q_l = 2^logql
 * (the rotation / conjugation permutations of src/poly.c:263-283 are the caller's). */
size_t gpq_he_swk_workspace_bytes(gpq_ctx *ctx, unsigned W, unsigned dimB, unsigned dimP, unsigned batch);
int gpq_he_swk(gpq_ctx *ctx, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *d0, const uint64_t *d1,
               const uint64_t *swk0, const uint64_t *swk1, unsigned W, unsigned logql, unsigned dimB, unsigned dimP,
               unsigned batch, void *workspace, void *stream);

/* he_mulpt, src/he-mult.c:159-196 (decl src/gpqhe.h:148): ciphertext times plaintext polynomial m on big slabs,
 * q_l = 2^logql, over `dim` limbs (the reference derives dim from log2(pt->nu), :169: caller's). */
size_t gpq_he_mulpt_workspace_bytes(const gpq_ctx *ctx, unsigned dim, unsigned batch);
int gpq_he_mulpt(gpq_ctx *ctx, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *c0, const uint64_t *c1, const uint64_t *m,
                 unsigned W, unsigned logql, unsigned dim, unsigned batch, void *workspace, void *stream);
/* poly_rot / poly_conj, src/poly.c:263-283, on big slabs (r must not alias a): with gpq_he_swk they are he_rot / he_conj,
 * src/he-automorphism.c:87-115. */
int gpq_poly_rot(gpq_ctx *ctx, uint64_t *r, const uint64_t *a, unsigned W, unsigned rot, unsigned batch, void *stream);
int gpq_poly_conj(gpq_ctx *ctx, uint64_t *r, const uint64_t *a, unsigned W, unsigned batch, void *stream);

/* ---- general moduli: any q_l (little-endian words ql_words[0..Lq)) and any Delta (uint64_t, as hectx_init takes it,
 * src/gpqhe.h:100).  Same reference semantics, through the multiword Barrett kernel: slow-path quality, meant for parameter
 * sets outside the powers of two that the fast entry points above cover. */
size_t gpq_he_general_workspace_bytes(gpq_ctx *ctx, unsigned W, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch);
int gpq_he_rs_general(gpq_ctx *ctx, uint64_t *c0, uint64_t *c1, unsigned W, unsigned long long delta, const uint64_t *ql_words,
                      unsigned Lq, unsigned batch, void *scratch /* 192 words */, void *stream);
int gpq_relin_tail_general(gpq_ctx *ctx, uint64_t *out, const uint64_t *chat, const uint64_t *d, unsigned W, const uint64_t *ql_words,
                           unsigned Lq, unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream);
int gpq_he_mul_general(gpq_ctx *ctx, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *ct1c0, const uint64_t *ct1c1,
                       const uint64_t *ct2c0, const uint64_t *ct2c1, const uint64_t *rlk0, const uint64_t *rlk1, unsigned W,
                       const uint64_t *ql_words, unsigned Lq, unsigned dimA, unsigned dimB, unsigned dimP, unsigned batch,
                       void *workspace, void *stream);
/* workspace: gpq_he_mulpt_workspace_bytes + gpq_poly_mul_general_workspace_bytes */
int gpq_he_mulpt_general(gpq_ctx *ctx, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *c0, const uint64_t *c1, const uint64_t *m,
                         unsigned W, const uint64_t *ql_words, unsigned Lq, unsigned dim, unsigned batch, void *workspace, void *stream);
int gpq_he_swk_general(gpq_ctx *ctx, uint64_t *out_c0, uint64_t *out_c1, const uint64_t *d0, const uint64_t *d1,
                       const uint64_t *swk0, const uint64_t *swk1, unsigned W, const uint64_t *ql_words, unsigned Lq,
                       unsigned dimB, unsigned dimP, unsigned batch, void *workspace, void *stream);

/* he_add / he_sub / he_neg on one big-slab polynomial (src/he-add.c:32-142: mpi_addm / mpi_subm + mpi_smod), q_l = 2^logql.
 * Not NTT work; offered so that ciphertexts can stay in HBM between multiplications.  r may alias a or b. */
int gpq_big_add(gpq_ctx *ctx, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned W, unsigned logql, unsigned batch, void *stream);
int gpq_big_sub(gpq_ctx *ctx, uint64_t *r, const uint64_t *a, const uint64_t *b, unsigned W, unsigned logql, unsigned batch, void *stream);
int gpq_big_neg(gpq_ctx *ctx, uint64_t *r, const uint64_t *a, unsigned W, unsigned logql, unsigned batch, void *stream);
/* The storage step of he_genswk (src/he-kem.c:103-110): a centred key polynomial (big slab) -> NTT-domain slab of dimevk limbs. */
int gpq_evk_pack(gpq_ctx *ctx, uint64_t *evk, const uint64_t *big, unsigned W, unsigned dimevk, unsigned batch, void *stream);
/* he_genswk, src/he-kem.c:74-118, from the polynomials the reference samples on the host: p1 (uniform mod P*q_L), e (error) and the
 * polynomial the key hides (sp = s^2 for he_genrlk, the rotated / conjugated secret for he_genrk / he_genck), all big slabs of W words
 * (64 W > bits of P*q_L), q_L = 2^logqL:  swk.p0 = smod(-p1*sk + e + P*sp, P*q_L), swk.p1 = smod(p1, P*q_L), each stored as
 * rns_decompose + ntt over dimevk limbs (evk0, evk1: uint64_t[dimevk][n]). */
size_t gpq_he_genswk_workspace_bytes(gpq_ctx *ctx, unsigned W, unsigned dimP, unsigned logqL);
int gpq_he_genswk(gpq_ctx *ctx, uint64_t *evk0, uint64_t *evk1, const uint64_t *p1, const uint64_t *sk, const uint64_t *e,
                  const uint64_t *sp, unsigned W, unsigned dimP, unsigned logqL, unsigned dimevk, void *workspace, void *stream);

/* ---- per-kernel profile ----------------------------------------------------
 * When enabled every kernel launch of this context is bracketed by two HIP
 * events on its own stream.  gpq_profile_collect waits for them and adds the
 * durations per kernel (index < gpq_profile_kernels()) into the two arrays. */
int gpq_profile_enable(gpq_ctx *ctx, int on);
int gpq_profile_kernels(void);
const char *gpq_profile_kernel_name(int kernel);
int gpq_profile_collect(gpq_ctx *ctx, double *total_ms, unsigned long long *launches);

/* ---- timing on the stream the kernels run on (HIP events) ----------------- */
typedef struct gpq_timer gpq_timer;
int gpq_timer_create(gpq_timer **t);
int gpq_timer_start(gpq_timer *t, void *stream);
int gpq_timer_stop(gpq_timer *t, void *stream);
int gpq_timer_elapsed_ms(gpq_timer *t, float *ms); /* synchronises on the stop event */
void gpq_timer_destroy(gpq_timer *t);

#ifdef __cplusplus
}
#endif
#endif /* GPQHE_HIP_H */
