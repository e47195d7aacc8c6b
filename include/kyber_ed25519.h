/*
 * kyber_ed25519.h — C ABI of the MI355X-native batched Ed25519 engine.
 *
 * Drop-in boundary for the `impl group::Point for Point` of teleconsys/kyber-rs
 * (/root/reference src/group/edwards25519/point.rs:75-225).  The reference has no FFI of its own;
 * these are the entry points its Rust side would bind (INTEGRATION.md shows the `extern "C"` block
 * and the trait impl that forwards to them).  Plain pointers and sizes only.
 *
 * Conventions
 *   scalar     32 bytes little-endian, exactly `Scalar.v` (scalar.rs:23-26).  NOT reduced or range
 *              checked: clamped EdDSA keys and scalars >= 2^255 behave as in ge.rs:442-568.
 *   enc        32-byte point encoding = `Point::marshal_binary` (point.rs:35-41, ge.rs:112-122).
 *   ext        40 x int32: X[10] Y[10] Z[10] T[10], radix-2^25.5 limbs = `Point.ge`
 *              (`ExtendedGroupElement`, ge.rs:78-83).  Inputs may be any limbs the reference itself
 *              produces (|limb| < 2^29 accepted).  Outputs are the limbs `fe_from_bytes`
 *              (fe.rs:67-122) yields for the canonical coordinate value: signed, centred,
 *              |even limb| <= 2^25, |odd limb| <= 2^24 — inside the reference's fe_mul/fe_add input
 *              bounds, so a returned point can be fed straight back into the CPU arithmetic.
 *   return     0 = ok; negative = KYB_E_* (nothing is written on error except where noted; a call that fails after it has queued part of
 *              its kernels — a device allocation or a launch refused half way — waits for the device before it returns, leaves the
 *              contents of its OUTPUT arrays undefined and the context usable: the next call gives the ordinary results).
 *   threading  every call may be made from any thread.  Calls act on the calling thread's CONTEXT (kyb_ctx_set_current;
 *              default = the context kyb_init made) and make that context's device current first.  Host-pointer calls
 *              on one context are serialised on its own streams (use one context per concurrent caller, or per GPU);
 *              device-pointer calls on different streams may overlap on the GPU — each stream gets its own scratch
 *              slot (up to 32 streams registered at a time; kyb_stream_release frees one, beyond 32 the least recently
 *              used slot is recycled after its last launch); their launch bookkeeping is serialised per context.
 *              kyb_set_option / kyb_profile_* are safe against concurrent launches.
 *              Concurrent small calls of several contexts need hardware queues: the ROCm runtime maps all streams of a process onto
 *              GPU_MAX_HW_QUEUES queues (default 4).  The variable is read when the HIP runtime starts and belongs to the host
 *              program: this library never touches the environment; a host that issues one-item calls from many threads sets
 *              GPU_MAX_HW_QUEUES=16 before its first HIP call (16 threads: 16k -> 72k variable-base calls/s).
 *   timing     The instruction stream, the LDS / memory addresses and the launch geometry of every kernel are independent of scalar
 *              VALUES, with documented exceptions, each a shape chosen from a PUBLIC property:
 *                ladder.skip_canonical (default 1): a launch whose scalars are ALL below 2^252 walks 252 bits instead of 256 — "is
 *                  reduced mod L", which holds for every scalar kyber-rs's Scalar type produces; set the option to 0 for 256 always;
 *                kyb_mul_public_batch / mul.short_scalars (default 0): multipliers DECLARED public by the caller skip their common
 *                  leading zero bits when all are below 2^64.  kyb_mul_batch itself never does.
 *              A third shape depends on how a POINT is written, never on what it is: kyb_encode_batch (and what marshals through it:
 *                  kyb_point_checks_batch on limbs) of at most four points per compute unit skips the field inversion for a point whose Z
 *                  limbs are literally (1, 0, ..., 0) — what unmarshal_binary yields and what this library returns unless ext.projective is
 *                  set; a projective result has that Z with probability 2^-255.
 *              tools/ct_check.py checks the claim on the compiled kernels (no branch on, and no address from, scalar words).
 *   memory     the caller owns every buffer; the library keeps no pointer after return.
 *   aliasing   host-pointer calls: an output array may be (or overlap) an input array of the same call — an in-place update such as
 *              out_ext == pts_ext — with the results of separate arrays: inputs are copied before anything is written, and page-locked
 *              caller arrays used where they lie (host.in_place) are staged instead when they share bytes with an array that is written.
 *              Two OUTPUT arrays of one call must not overlap.  Device-pointer (_dev) calls: no array may overlap another.
 *   secrets    host-pointer calls that take secret scalars (kyb_mul_base_batch, kyb_mul_batch, the signing calls, kyb_pripoly_eval_batch)
 *              clear, before they return and on every error path, the copies of the secret INPUTS the engine made — page-locked zero-copy
 *              buffer, device staging, bounce ring — and kyb_mul_batch (whose result s*P is a Diffie-Hellman shared secret) also the
 *              copies of its OUTPUTS in the zero-copy buffer and the device staging.  Not cleared per call: the per-stream device scratch
 *              (projective staging records; overwritten by the next call on the stream, wiped when the context or the stream slot is
 *              released) and, for pageable callers of calls of 2^16 items and more, the page-locked landing area of the results (buffers
 *              from kyb_host_alloc bypass it).  Device-pointer calls work on the caller's own buffers and leave only the scratch.
 *
 * Two flavours of every batch call:
 *   kyb_xxx_batch      host pointers; H2D copy, kernel, D2H copy, synchronous.
 *   kyb_xxx_batch_dev  device pointers (HBM-resident batches); asynchronous on `stream`
 *                      (a hipStream_t passed as void*, NULL = the engine's own stream).
 *                      ORDERING IS THE CALLER'S: the kernels are ordered with whatever else is queued on `stream` and with nothing
 *                      else.  The engine's own stream is created hipStreamNonBlocking: it does NOT wait for the device's null stream
 *                      (nor the null stream for it), so a caller that passes NULL must have FINISHED producing the operands
 *                      (hipStreamSynchronize / an event it waited for) before the call, and must kyb_sync(NULL) before it reads
 *                      the results or overwrites / frees any operand.  A caller whose producers and consumers run on a stream
 *                      passes THAT stream; for the null stream, whose handle is also NULL, pass KYB_STREAM_LEGACY.
 *
 * There is NO CPU implementation behind this ABI: without a usable gfx950 device kyb_init fails
 * with KYB_E_NO_DEVICE and every other call returns KYB_E_NOT_INIT.
 */
#ifndef KYBER_ED25519_H
#define KYBER_ED25519_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* `stream` argument of the _dev calls, kyb_sync and kyb_stream_release: the device's null (legacy default) stream */
#define KYB_STREAM_LEGACY ((void*)1)

#define KYB_OK 0
#define KYB_E_NOT_INIT (-1)
#define KYB_E_BAD_ARG (-2)
#define KYB_E_NO_DEVICE (-3)
#define KYB_E_HIP (-4)
#define KYB_E_NOMEM (-5)
#define KYB_E_TRANSPORT (-6)  /* kyb_group_create_ex(KYB_GROUP_REQUIRE_RCCL): the RCCL broadcast of the table image could not be done */
#define KYB_E_STALE (-7)      /* deferred points: the handle names a node the arena no longer has (see "Lifetime" at kyb_defer_*) */

/* table image exchanged between GPUs at init — the role of constants.rs:89 BASE (which holds the 32 even
 * radix-16 positions only); affine (y+x, y-x, 2dxy), canonical limbs:
 *   bytes [0, 65536)        uint32 [64 pos][8 quads][ 8 entries][4]   entry (pos, j) = (j+1) * 16^pos * B
 *   bytes [65536, 172032)   uint32 [52 pos][8 quads][16 entries][4]   entry (pos, j) = (j+1) * 32^pos * B
 *   bytes [172032, 335232)  42 windows of 3840 B with E = 32 entries, then a top window of 1920 B with E = 16:
 *                           per window the planes  y+x [2][E][4] | y-x [2][E][4] | y+x [E][2] | y-x [E][2] |
 *                           2dxy [2][E][4] | 2dxy [E][2]  (limbs 0..7 in the quad planes, 8..9 in the pair planes)
 *                                                                     entry (pos, j) = (2j+1) * 64^pos * B */
#define KYB_BASE_TABLE_BYTES 335232u

/* Version of this interface: bumped on any change of an existing signature, of KYB_BASE_TABLE_BYTES or of a status /
 * error code (a binding checks it once after loading the library).  Callable before kyb_init. */
#define KYB_ABI_VERSION 2
int kyb_abi_version(void);

/* ---- lifecycle -------------------------------------------------------------------------------- */
/* Create this process's DEFAULT context on HIP device `device`: streams, table image built on the GPU.  Idempotent for
 * the same device; a second device is refused here — more GPUs from one process go through kyb_ctx_create /
 * kyb_group_create below (or one process per GPU, as bench.py does: DESIGN.md §multi-GPU). */
int kyb_init(int device);
/* Same, but leave the base table empty: the caller fills it with kyb_base_table_import_dev after
 * receiving rank 0's image over RCCL (torch.distributed.broadcast). */
int kyb_init_no_table(int device);
void kyb_shutdown(void);           /* destroys the default context */
/* human-readable text of the last failure on the calling thread ("" if none) */
const char* kyb_last_error(void);
/* device name, CU count, bytes of workspace; any pointer may be NULL */
int kyb_device_info(char* name, size_t name_cap, int* compute_units, size_t* workspace_bytes);
/* block until everything queued on `stream` (NULL = engine stream) has finished */
int kyb_sync(void* stream);
/* Give back the scratch the context keeps for a caller stream (staging buffers, side stream, events).  Call it before
 * destroying a stream that carried kyb_*_dev calls; waits for the stream's last engine launch.  Unknown streams are
 * ignored. */
int kyb_stream_release(void* stream);

/* ---- explicit contexts: several GPUs (or several independent pipelines on one GPU) in one process ---------------- */
/* A context owns everything the calls need on one device: two streams, per-stream scratch, staging and bounce buffers,
 * the table image, options, profiling state.  Every kyb_* call acts on the calling thread's current context.
 *   kyb_ctx_create       new context on `device` (build_table = 0: import an image later); any number per device
 *   kyb_ctx_set_current  bind it to the calling thread (NULL = back to the default context of kyb_init)
 *   kyb_ctx_destroy      no thread may still be using it */
typedef struct kyb_ctx kyb_ctx;
int kyb_ctx_create(int device, int build_table, kyb_ctx** out);
int kyb_ctx_destroy(kyb_ctx* ctx);
int kyb_ctx_set_current(kyb_ctx* ctx);
kyb_ctx* kyb_ctx_get_current(void);
int kyb_ctx_device(const kyb_ctx* ctx);

/* ---- groups: SURVEY.md §8(e) in one process ------------------------------------------------------------------- */
/* kyb_group_create(devices, n): one context and (per call) one host thread per listed device.  The table image is built
 * on devices[0] and moved to the others with one ncclBroadcast (librccl over xGMI, loaded on demand; falls back to a
 * host copy when RCCL is unavailable or the list repeats a device) and validated against its embedded checksum.
 * kyb_group_table_transport reports which: "rccl", "host-copy" or "none" (one rank).
 * The kyb_group_*_batch calls shard a host-pointer batch: rank r takes items [floor(n r / G), floor(n (r+1) / G)) of
 * every array and runs the ordinary call on its own device; there is no data-path collective.  For device-resident
 * shards use kyb_group_ctx(g, r) + kyb_ctx_set_current and the kyb_*_dev calls from one thread per rank. */
typedef struct kyb_group kyb_group;
int kyb_group_create(const int* devices, int n, kyb_group** out);
/* The same with flags.  A fallback to the host copy is a success, but never a silent one: kyb_group_table_transport_note (and
 * kyb_last_error, until the next failure) say at which step the RCCL path stopped — library not loadable, a symbol missing,
 * ncclCommInitAll / ncclBroadcast returning an error, a repeated device.
 *   KYB_GROUP_REQUIRE_RCCL           no fallback: the call fails with KYB_E_TRANSPORT when the RCCL broadcast cannot be done (what a
 *                                    multi-GPU node's bring-up wants: `bench.py --mode group` sets it when the devices are distinct)
 *   KYB_GROUP_RCCL_EVEN_IF_REPEATED  attempt the RCCL transport although the list repeats a device.  Real RCCL refuses such a list; the flag
 *                                    exists so that a one-GPU box can drive the library's RCCL call sequence against a stand-in librccl
 *                                    (tests/test_gpu_rccl_stub.py). */
#define KYB_GROUP_REQUIRE_RCCL 1u
#define KYB_GROUP_RCCL_EVEN_IF_REPEATED 2u
int kyb_group_create_ex(const int* devices, int n, unsigned flags, kyb_group** out);
const char* kyb_group_table_transport_note(const kyb_group* g);
void kyb_group_destroy(kyb_group* g);
int kyb_group_size(const kyb_group* g);
kyb_ctx* kyb_group_ctx(kyb_group* g, int rank);
const char* kyb_group_table_transport(const kyb_group* g);
int kyb_group_mul_base_batch(kyb_group* g, const uint8_t* scalars, size_t n, uint8_t* out_enc, int32_t* out_ext);
int kyb_group_mul_batch(kyb_group* g, const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, size_t n,
                        uint8_t* out_enc, int32_t* out_ext, uint8_t* ok);
int kyb_group_schnorr_sign_batch(kyb_group* g, const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* msg_off,
                                 size_t n, uint8_t* sig);
int kyb_group_verify_batch(kyb_group* g, const uint8_t* pubs, const uint8_t* msgs, const uint32_t* msg_off, const uint8_t* sigs,
                           size_t n, int flavor, uint8_t* status);
/* Device-resident shards: argument r of every array is rank r's pointer / item count, the memory on rank r's GPU (16-byte aligned,
 * as for the kyb_*_dev calls).  The launches are asynchronous on each context's own stream — one calling thread queues the work of
 * all GPUs — and kyb_group_sync waits for everything queued on every rank.  Those streams are non-blocking (ordered with no other
 * stream of the process): the caller has FINISHED producing the shards before the call and reads results after kyb_group_sync.
 * Arrays that are NULL as a whole are absent for all ranks (out_ext, ok; exactly one of pts_enc / pts_ext). */
int kyb_group_mul_base_batch_dev(kyb_group* g, const uint8_t* const* scalars, const size_t* n, uint8_t* const* out_enc, int32_t* const* out_ext);
int kyb_group_mul_batch_dev(kyb_group* g, const uint8_t* const* scalars, const uint8_t* const* pts_enc, const int32_t* const* pts_ext,
                            const size_t* n, uint8_t* const* out_enc, int32_t* const* out_ext, uint8_t* const* ok);
int kyb_group_sync(kyb_group* g);

/* ---- pinned host memory for the host-pointer API ------------------------------------------------ */
/* The host-pointer calls accept any host memory.  From pageable memory the copies run at ~7 GB/s and
 * dominate (2^20 variable-base items: 33 ms against 9.4 ms of kernels); batch buffers obtained here are
 * page-locked, DMA at PCIe rate and let the chunked two-stream pipeline overlap copies with kernels.
 * The memory is host-coherent and mapped for the context's device: a call small enough to run without copies
 * (option host.zero_copy_kib) reads and writes arrays that lie in it where they lie (option host.in_place).
 * Returns NULL on failure (kyb_last_error).  Free with kyb_host_free. */
void* kyb_host_alloc(size_t bytes);
void kyb_host_free(void* p);

/* ---- base-point table (multi-GPU init) -------------------------------------------------------- */
/* The image carries a 64-bit checksum of itself (in the two padding words of radix-16 entry (0, 0)); both import calls
 * recompute it on the GPU and refuse an image that does not match (KYB_E_BAD_ARG, the table stays unusable): a truncated
 * or corrupted broadcast cannot silently produce wrong points. */
int kyb_base_table_export_dev(void* dst_dev, void* stream);        /* engine table -> dst (device)  */
int kyb_base_table_import_dev(const void* src_dev, void* stream);  /* src (device) -> engine table  */
int kyb_base_table_export(uint8_t* dst_host);                      /* engine table -> host buffer   */
int kyb_base_table_import(const uint8_t* src_host);                /* host buffer -> engine table   */

/* ---- Point::mul(s, None): fixed base — ge_scalar_mult_base, ge.rs:442-486 --------------------- */
/* out_enc (n x 32) and/or out_ext (n x 40 int32, Z = 1 unless the option ext.projective is set) may be NULL, not both. */
int kyb_mul_base_batch(const uint8_t* scalars, size_t n, uint8_t* out_enc, int32_t* out_ext);
int kyb_mul_base_batch_dev(const uint8_t* scalars, size_t n, uint8_t* out_enc, int32_t* out_ext, void* stream);

/* ---- Point::mul(s, Some(P)): variable base — ge_scalar_mult, ge.rs:508-568 -------------------- */
/* Exactly one of pts_enc (n x 32, decoded on the GPU as unmarshal_binary does) / pts_ext (n x 40).
 * pts_ext must hold points ON the curve (what kyb_decode_batch, any kyb_* output or the reference's own arithmetic
 * produce): the default kernel is an x-only Montgomery ladder that never reads T and relies on the curve equation, so
 * for off-curve limbs — where the reference's formulas return some deterministic garbage — the result here is
 * unspecified (the other items of the batch are unaffected).
 * ok (n bytes, may be NULL unless pts_enc is given): 1 = point decoded, 0 = invalid encoding (the
 * outputs of that item are then the encoding of the neutral element / unspecified limbs). */
int kyb_mul_batch(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, size_t n,
                  uint8_t* out_enc, int32_t* out_ext, uint8_t* ok);
int kyb_mul_batch_dev(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, size_t n,
                      uint8_t* out_enc, int32_t* out_ext, uint8_t* ok, void* stream);
/* The same multiplication for multipliers the caller DECLARES public — share indices x = i + 1 of PubPoly::eval (poly.rs:461-464),
 * the cofactor of Point::pick (point.rs:148), Lagrange coefficients of public indices.  A call of at most 64 items (pts_ext) whose
 * scalars are ALL below 2^64 starts its ladder below their common leading zero bits: 29 us instead of 158 for a 10-bit index.  The
 * run time therefore depends on the scalars' bit length: never pass a secret here.  Same results as kyb_mul_batch. */
int kyb_mul_public_batch(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, size_t n,
                         uint8_t* out_enc, int32_t* out_ext, uint8_t* ok);

/* ---- Point::add / Point::sub, point.rs:179-197 (out = a +/- b, extended limbs, Z arbitrary) ---- */
int kyb_add_batch(const int32_t* a_ext, const int32_t* b_ext, size_t n, int32_t* out_ext, int subtract);
int kyb_add_batch_dev(const int32_t* a_ext, const int32_t* b_ext, size_t n, int32_t* out_ext, int subtract, void* stream);

/* ---- marshal_binary / unmarshal_binary, point.rs:35-51 ---------------------------------------- */
/* encode: one field inversion per 8 points (Montgomery's trick, k_encode_batched) instead of the reference's one per
 * point (ge.rs:112-122); the encodings are identical. */
int kyb_encode_batch(const int32_t* pts_ext, size_t n, uint8_t* out_enc);
int kyb_encode_batch_dev(const int32_t* pts_ext, size_t n, uint8_t* out_enc, void* stream);
/* ok[i] = 1 iff enc[i] decodes (ge.rs:124-179: non-canonical y accepted, x = 0 with sign accepted) */
int kyb_decode_batch(const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok);
int kyb_decode_batch_dev(const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok, void* stream);

/* ---- schnorr::sign with caller-supplied nonce, schnorr_sig.rs:25-47 ---------------------------- */
/* x, k: n x 32 scalars; msgs: concatenated messages, message i = msgs[msg_off[i] .. msg_off[i+1])
 * (msg_off has n+1 entries); sig: n x 64 = enc(k*B) || (k + x*h mod L). */
int kyb_schnorr_sign_batch(const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* msg_off,
                           size_t n, uint8_t* sig);
int kyb_schnorr_sign_batch_dev(const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* msg_off,
                               size_t n, uint8_t* sig, void* stream);
/* The same for callers that hold the public keys (DSS long-term keys, dss_sig.rs:215-237; EdDSA objects):
 * pubs = n x 32 encodings of x*B, hashed as given, so A is not recomputed and a signature costs one
 * fixed-base multiplication instead of two.  pubs == NULL behaves like kyb_schnorr_sign_batch.  A pubs[i]
 * that is not enc(x[i]*B) yields a signature that does not verify (as it would in the reference if a key
 * object held a wrong public key). */
int kyb_schnorr_sign_keyed_batch(const uint8_t* x, const uint8_t* pubs, const uint8_t* k, const uint8_t* msgs,
                                 const uint32_t* msg_off, size_t n, uint8_t* sig);
int kyb_schnorr_sign_keyed_batch_dev(const uint8_t* x, const uint8_t* pubs, const uint8_t* k, const uint8_t* msgs,
                                     const uint32_t* msg_off, size_t n, uint8_t* sig, void* stream);

/* ---- EdDSA::sign, eddsa_sig.rs:120-152 with the key expansion of curve.rs:74-87 ------------------ */
/* seeds: n x 32.  Per item: d = SHA-512(seed), x = clamp(d[0..32)) (unreduced), prefix = d[32..64),
 * r = SHA-512(prefix || msg) mod L, then the Schnorr equations with nonce r: sig = enc(r*B) || (r + x*h mod L),
 * h = SHA-512(enc R || enc A || msg) mod L, A = x*B.  pub (n x 32, may be NULL) receives enc(A), the public key
 * (`EdDSA::new`, eddsa_sig.rs:31-43).  This is the computation the reference's golden file pins. */
int kyb_eddsa_sign_batch(const uint8_t* seeds, const uint8_t* msgs, const uint32_t* msg_off, size_t n, uint8_t* sig, uint8_t* pub);
int kyb_eddsa_sign_batch_dev(const uint8_t* seeds, const uint8_t* msgs, const uint32_t* msg_off, size_t n, uint8_t* sig, uint8_t* pub, void* stream);
/* EdDSA::sign on existing key objects: the struct already holds the public key (eddsa_sig.rs:22-29, used at
 * :132-137), so only R = r*B is computed.  pubs = n x 32, required. */
int kyb_eddsa_sign_keyed_batch(const uint8_t* seeds, const uint8_t* pubs, const uint8_t* msgs, const uint32_t* msg_off, size_t n, uint8_t* sig);
int kyb_eddsa_sign_keyed_batch_dev(const uint8_t* seeds, const uint8_t* pubs, const uint8_t* msgs, const uint32_t* msg_off, size_t n, uint8_t* sig, void* stream);

/* ---- signature verification with the reference's checks --------------------- */
/* eddsa::verify_with_checks (eddsa_sig.rs:159-212, flavor 0) / schnorr::verify_with_checks
 * (schnorr_sig.rs:53-110, flavor 1): s*B == R + SHA-512(R || A || msg)*A after the canonical and
 * small-order checks; the flavors differ only in which error wins when several apply.
 * pubs: n x 32, sigs: n x 64 (R || s), msgs/msg_off as for signing.  status[i]: 0 valid,
 * 2 SignatureNotCanonical, 3 RNotCanonical, 4 R does not decode, 5 RSmallOrder,
 * 6 PublicKeyNotCanonical, 7 public key does not decode, 8 PublicKeySmallOrder, 9 InvalidSignature
 * (1, wrong signature length, cannot occur with fixed 64-byte records). */
int kyb_verify_batch(const uint8_t* pubs, const uint8_t* msgs, const uint32_t* msg_off, const uint8_t* sigs,
                     size_t n, int flavor, uint8_t* status);
int kyb_verify_batch_dev(const uint8_t* pubs, const uint8_t* msgs, const uint32_t* msg_off, const uint8_t* sigs,
                         size_t n, int flavor, uint8_t* status, void* stream);
/* The same verification with the public keys given as POINTS (160-byte extended limbs, any Z): schnorr::verify (schnorr_sig.rs:114-127) and
 * eddsa::verify (eddsa_sig.rs:216-...) — what dkg.rs, vss.rs and dss_sig.rs call — take &Point, marshal it and hand the bytes to
 * verify_with_checks, which unmarshals them again.  Here the engine marshals the points (one shared inversion per 8) for the hash and the
 * byte-level checks and multiplies the caller's limbs directly: no square root per key unless a set of limbs is not a point of the curve, in
 * which case its bytes decide, as in the reference.  status[i] == kyb_verify_batch on marshal_binary(point i), for every input. */
int kyb_verify_points_batch(const int32_t* pubs_ext, const uint8_t* msgs, const uint32_t* msg_off, const uint8_t* sigs, size_t n, int flavor, uint8_t* status);
int kyb_verify_points_batch_dev(const int32_t* pubs_ext, const uint8_t* msgs, const uint32_t* msg_off, const uint8_t* sigs, size_t n, int flavor,
                                uint8_t* status, void* stream);

/* ---- PubPoly::eval / shares, poly.rs:457-478 ------------------------------- */
/* One public polynomial (t commitments, extended limbs) evaluated at n share indices:
 * out[i] = sum_j commits[j] * (indices[i] + 1)^j, exactly what eval(indices[i]) returns.  The share
 * index is public, so the small scalar x = index + 1 is handled by a short binary ladder instead of the
 * reference's 64-window multiplication; the resulting point (hence its encoding) is the same.
 * max_index (dev flavour) = an upper bound of indices[], it sets the ladder length. */
int kyb_pubpoly_eval_batch(const int32_t* commits_ext, size_t t, const uint32_t* indices, size_t n,
                           uint8_t* out_enc, int32_t* out_ext);
int kyb_pubpoly_eval_batch_dev(const int32_t* commits_ext, size_t t, const uint32_t* indices, size_t n, uint32_t max_index,
                               uint8_t* out_enc, int32_t* out_ext, void* stream);

/* m public polynomials of the same threshold t (commits_ext = m x t x 40 int32), each evaluated at its own k
 * indices: out[g*k + j] = polynomial g at indices[g*k + j].  k = 1 is the verifier's side of a DKG round: the deals
 * of m dealers checked at the verifier's own index (vss/pedersen/vss.rs:904-909) in one launch. */
int kyb_pubpoly_eval_multi_batch(const int32_t* commits_ext, size_t t, size_t m, const uint32_t* indices, size_t k,
                                 uint8_t* out_enc, int32_t* out_ext);
int kyb_pubpoly_eval_multi_batch_dev(const int32_t* commits_ext, size_t t, size_t m, const uint32_t* indices, size_t k, uint32_t max_index,
                                     uint8_t* out_enc, int32_t* out_ext, void* stream);

/* ---- recover_commit / recover_pub_poly / PubPoly::add, poly.rs:486-507, 566-634 -- */
/* m linear combinations of t points each:  out[g] = sum_{j<t} scalars[g*t + j] * P(g, j).
 *   shared_points == 0:  P(g, j) = pts[g*t + j]   (m*t points)   - m independent recover_commit calls
 *                        (scalars = the Lagrange coefficients num/den of poly.rs:580-594, computed by the caller)
 *   shared_points == 1:  P(g, j) = pts[j]         (t points)     - recover_pub_poly: coefficient g of
 *                        sum_j L_j * y_j with scalars[g*t + j] = coefficient g of the Lagrange basis L_j
 * Exactly one of pts_enc (32-byte encodings, decoded like kyb_mul_batch; `ok` gets one flag per POINT,
 * failed decodes count as the neutral element) / pts_ext (reference limbs).  The reference performs the t
 * multiplications and t additions one by one; here: one ladder launch over the m*t products and
 * ceil(log2 t) pairwise-addition passes in HBM.  The sum is a group element, so its encoding does not
 * depend on the order of the additions.  t = 1 degenerates to kyb_mul_batch. */
int kyb_lincomb_batch(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, int shared_points,
                      size_t m, size_t t, uint8_t* out_enc, int32_t* out_ext, uint8_t* ok);
int kyb_lincomb_batch_dev(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, int shared_points,
                          size_t m, size_t t, uint8_t* out_enc, int32_t* out_ext, uint8_t* ok, void* stream);

/* The same linear combinations for scalars the CALLER DECLARES PUBLIC — the Lagrange basis coefficients of recover_pub_poly
 * (poly.rs:607-668) and the Lagrange coefficients of recover_commit (poly.rs:580-594) are functions of public share indices.  With
 * shared_points = 1 and enough outputs to pay for it (m >= 16, m * t >= 8192, t <= 8192) every point gets a table of its radix-64 window
 * multiples on the GPU and a product costs 43 table additions instead of a 255-step ladder (t = m = 683: several times faster); table
 * addresses and skipped zero digits depend on the scalars, so NEVER pass a secret here.  Every other shape takes kyb_lincomb_batch's
 * constant-time path.  Same results as kyb_lincomb_batch, also on small-order and mixed-order points. */
int kyb_lincomb_public_batch(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, int shared_points,
                             size_t m, size_t t, uint8_t* out_enc, int32_t* out_ext, uint8_t* ok);
int kyb_lincomb_public_batch_dev(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, int shared_points,
                                 size_t m, size_t t, uint8_t* out_enc, int32_t* out_ext, uint8_t* ok, void* stream);

/* PriPoly::eval / PriPoly::shares — src/share/poly.rs:133-152: out_shares[g][i] = sum_j coeffs[g][j] * x_i^j mod L with x_i = indices[i] + 1,
 * for m SECRET polynomials of t coefficients (32-byte scalars, any 256-bit value, used as the reference's sc_mul / sc_add use them) at k
 * public share indices; outputs are the canonical residues the reference's Scalar holds.  A dealer's n shares of a threshold-t polynomial
 * are n * t scalar multiply-adds on one core in the reference; here the Horner chains are cut into segments, a lane each.  Constant time in
 * the coefficients (the arithmetic of the signing kernel; loop bounds from t and the public indices only); host staging is wiped. */
int kyb_pripoly_eval_batch(const uint8_t* coeffs, size_t m, size_t t, const uint32_t* indices, size_t k, uint8_t* out_shares);
int kyb_pripoly_eval_batch_dev(const uint8_t* coeffs, size_t m, size_t t, const uint32_t* indices, size_t k, uint8_t* out_shares, void* stream);

/* The scalar side of recover_commit (poly.rs:580-594): for m share sets of t PUBLIC share indices each,
 *     out[g*t + i] = prod_{j != i} x_j / (x_j - x_i) mod L,   x = index + 1,
 * the Lagrange coefficients at 0, as canonical 32-byte scalars — the reference's num / den with its Scalar::div (= multiplication by
 * den^(L-2), scalar.rs:185-215).  The reference walks the t^2 products and t inversions on one core; here one lane per coefficient.
 * Indices of one set must be distinct (the reference keys its shares by index); a repeated index gives 0 for the coefficients it touches.
 * Feed the result to kyb_lincomb_public_batch (shared_points = 0: every set has its own share points). */
int kyb_lagrange_coeffs_batch(const uint32_t* indices, size_t m, size_t t, uint8_t* out_scalars);
int kyb_lagrange_coeffs_batch_dev(const uint32_t* indices, size_t m, size_t t, uint8_t* out_scalars, void* stream);

/* Sums of points without scalars: out[g] = sum_{j<t} pts[g*t + j] (m groups of t points, extended limbs).  The
 * distributed public polynomial of a DKG round is the coefficient-wise sum of the dealers' commitment polynomials
 * (dkg.rs:905-953 applies PubPoly::add, poly.rs:486-507, dealer after dealer): m = threshold, t = number of dealers,
 * with the commitments laid out coefficient-major. */
int kyb_sum_batch(const int32_t* pts_ext, size_t m, size_t t, uint8_t* out_enc, int32_t* out_ext);
int kyb_sum_batch_dev(const int32_t* pts_ext, size_t m, size_t t, uint8_t* out_enc, int32_t* out_ext, void* stream);

/* ---- the same two, from the wire format (the data format on the caller's side of N1) ---- */
/* A Deal carries its dealer's commitments as 32-byte encodings (vss/pedersen/vss.rs:113-124, Deal.commitments; the
 * reference unmarshals them one by one, point.rs:43-51, before PubPoly::eval / PubPoly::add see them).  These entry
 * points take the encodings as they arrive and decode on the GPU: 32 instead of 160 bytes per commitment over PCIe
 * and no host-side decode.  ok (may be NULL) gets one flag per encoding; one that does not decode counts as the
 * neutral element, as in kyb_lincomb_batch — the caller rejects that dealer, as the reference's unmarshal error does.
 *   kyb_pubpoly_eval_multi_enc_batch: commits_enc = m x t x 32 bytes, polynomial g = encodings [g*t, (g+1)*t).
 *   kyb_sum_enc_batch: item_major == 0: out[g] = sum_j pts[g*t + j]  (as kyb_sum_batch)
 *                      item_major != 0: out[g] = sum_j pts[j*m + g]  — t dealers' polynomials of m coefficients each, in
 *                      the order they were received; no transposition on the host. */
int kyb_pubpoly_eval_multi_enc_batch(const uint8_t* commits_enc, size_t t, size_t m, const uint32_t* indices, size_t k,
                                     uint8_t* out_enc, int32_t* out_ext, uint8_t* ok);
int kyb_pubpoly_eval_multi_enc_batch_dev(const uint8_t* commits_enc, size_t t, size_t m, const uint32_t* indices, size_t k, uint32_t max_index,
                                         uint8_t* out_enc, int32_t* out_ext, uint8_t* ok, void* stream);
int kyb_sum_enc_batch(const uint8_t* pts_enc, size_t m, size_t t, int item_major, uint8_t* out_enc, int32_t* out_ext, uint8_t* ok);
int kyb_sum_enc_batch_dev(const uint8_t* pts_enc, size_t m, size_t t, int item_major, uint8_t* out_enc, int32_t* out_ext, uint8_t* ok, void* stream);

/* ---- the verifier's side of one DKG round in ONE call ---------------------------------------------------------------------- */
/* Every node of a Pedersen DKG receives m deals, each with its dealer's t commitments as 32-byte encodings (Deal.commitments,
 * vss/pedersen/vss.rs:113-124), and needs two things from the same m x t points: dealer g's public polynomial evaluated at the node's own
 * share index (vss.rs:904-909, PubPoly::eval, poly.rs:457-469) and the coefficient-wise sum over the dealers (the distributed public
 * polynomial, dkg.rs:905-953 folding PubPoly::add, poly.rs:486-507).  kyb_pubpoly_eval_multi_enc_batch + kyb_sum_enc_batch do that
 * in two calls that each move and decode the m x t encodings; this one moves and decodes them once.
 *   commits_enc  m x t x 32, dealer after dealer, as received          index     the node's share index (x = index + 1)
 *   eval_enc / eval_ext   m outputs: polynomial g at `index` (one of the two may be NULL)
 *   sum_enc / sum_ext     t outputs: sum over the dealers of commitment j (both NULL: evaluations only)
 *   ok           m x t flags (may be NULL): 0 = that encoding is not a point; it then counts as the neutral element in both results,
 *                and the caller rejects the dealer as the reference's unmarshal error does
 * index_dev (device flavour): m copies of `index` in device memory. */
int kyb_dkg_verify_round_enc(const uint8_t* commits_enc, size_t t, size_t m, uint32_t index, uint8_t* eval_enc, int32_t* eval_ext,
                             uint8_t* sum_enc, int32_t* sum_ext, uint8_t* ok);
int kyb_dkg_verify_round_enc_dev(const uint8_t* commits_enc, size_t t, size_t m, const uint32_t* index_dev, uint32_t index, uint8_t* eval_enc,
                                 int32_t* eval_ext, uint8_t* sum_enc, int32_t* sum_ext, uint8_t* ok, void* stream);

/* ---- Point::eq, point.rs:227-241 -------------------------------------------- */
/* eq[i] = 1 iff a[i] and b[i] have the same encoding (the reference compares the two encodings = two
 * inversions; here a projective cross-multiplication).  Records with Z = 0 (`Point::default()`) behave as in the
 * reference, where they encode as x = y = 0. */
int kyb_equal_batch(const int32_t* a_ext, const int32_t* b_ext, size_t n, uint8_t* eq);
int kyb_equal_batch_dev(const int32_t* a_ext, const int32_t* b_ext, size_t n, uint8_t* eq, void* stream);

/* ---- deferred points: batches for callers that stay element-at-a-time (group.rs:85-140) -------------------------------------- */
/* kyber-rs's protocol code calls Point::mul / add / sub / neg one element at a time and looks at the result later, when it marshals,
 * compares or hashes it; a batch-of-1 engine call costs more than the CPU's own multiplication.  With these calls a binding hands the
 * operations over WITHOUT asking for results: every call records a node in the calling context's arena and returns a handle (a 64-bit
 * sequence number, never reused; 0 is no handle); nothing runs until bytes or limbs are asked for.  A flush evaluates the recorded
 * graph with as few engine calls as it can (csrc/defer.inc):
 *   - nodes whose operands are known go out grouped by operation, one batch call per group and dependency level
 *     (PriPoly::commit, poly.rs:195-206: t independent multiplications = ONE kyb_mul_batch / kyb_mul_base_batch call);
 *   - a Horner chain v = x v + c_j with one multiplier 1 <= x < 2^32 (PubPoly::eval, poly.rs:457-469: 2 t dependent calls) becomes ONE
 *     kyb_pubpoly_eval_multi_batch call, shared by all chains of that length in the flush.  The chain's multiplier is thereby treated as
 *     PUBLIC (it is a share index; a uniformly random secret has that shape with probability 2^-220);
 *   - a chain of additions acc = acc + term_j (recover_commit, poly.rs:566-603; the sums of dkg.rs:905-953): the terms in one batch,
 *     then ONE kyb_sum_batch call;
 *   - when a flush is caused by a request for BYTES, everything it evaluates is marshalled by the same kernels (their shared
 *     inversions) and cached: the marshal_binary calls that follow cost no engine call.
 * Results: the same group elements, hence the same 32 bytes, as the eager calls; the limbs are one valid representation of the point
 * (as everywhere in this ABI).  Option defer.fuse = 0 switches the chain recognition off (level-by-level batches only).
 * Lifetime: the reference's Point is Copy, so copies of a handle may live anywhere — in protocol state for the life of a node (dkg.rs:41,170;
 * dss_sig.rs:44) — and nodes are not reference-counted.  The arena therefore keeps two things:
 *   the WINDOW   the youngest defer.max_nodes nodes (default 2^18) with their graph: 40 bytes a node, plus 224 where it holds a scalar or a
 *                value — at most 69 MB.  When it is full it moves on by a quarter: first everything still pending is evaluated (one flush),
 *                then the oldest quarter leaves;
 *   KEPT VALUES  a node that leaves the window WITH a value leaves its 160 bytes of limbs (and its 32 bytes, if it has them) in a table keyed
 *                by its handle: the handle keeps working — asked for, it answers from the table; used as an operand, it comes back into the
 *                window as a leaf.  The table is bounded by defer.keep_mib (default 256 MiB = 1.29 million values of 208 bytes, plus about a
 *                third for its index); when full, the values nobody has touched since they were last looked over go first (second-chance
 *                order), so a distributed public key that is marshalled or multiplied every round stays.
 * A handle is refused with KYB_E_STALE ("stale handle"), never answered wrongly, only when (1) kyb_defer_floor(mark) dropped it — everything
 * recorded before `mark` (= an earlier kyb_defer_mark()), values included: the host's own statement that a round is over; (2) its node never
 * had a value when the window left it — the inner steps of a chain that was evaluated as ONE call, which exist as locals that the reference's
 * loops overwrite (poly.rs:457-469, 566-603); (3) its value was pushed out of the table: untouched while defer.keep_mib of younger values
 * arrived (at 64 participants a Pedersen dealer round leaves 300 to 750 values: the table holds the last 1,700 to 4,600 rounds, what is
 * touched stays longer).  A binding whose point still
 * holds its limbs registers them again (host/edwards25519.hpp).  Secret scalars are kept until their node is evaluated and cleared then; a
 * node's limbs and bytes — a recorded Diffie-Hellman exchange leaves the shared point there — are cleared when they leave the window without
 * being kept, when they leave the table, and at the end of the arena (the evaluator's own staging arrays as they are released): call
 * kyb_defer_floor when a round's secrets are done with.
 * A handle carries the number of the arena it came from (upper 24 bits), so handles of two contexts never collide and a point recorded
 * through one context may be read, compared or used as an operand through another (the reference's Point is Send: a worker thread may hand
 * its points to the thread that marshals them) — the evaluation then runs on the reader's context.  The arena of a released context stays
 * readable as an orphan until sixteen younger orphans exist.
 *   kyb_defer_input   a point the caller holds (40 limbs) as a leaf          kyb_defer_null / _base   the neutral element / B
 *   kyb_defer_input_enc   the same, with the 32 bytes marshal_binary yields for it when the caller knows them (NULL: unknown) — a point that was
 *                     unmarshalled from its canonical encoding: the leaf has its bytes from the start, so comparing it with an evaluated point
 *                     (the search of the own key among the participants' keys, dss_sig.rs:180-190) is 32 bytes against 32 bytes, no engine call.
 *                     The bytes are taken on trust: they must be what kyb_encode_batch would return for these limbs.
 *   kyb_defer_mul_base / _mul / _add (subtract != 0: a - b) / _neg            Point::mul(s, None) / mul(s, Some(p)) / add / sub / neg
 *   kyb_defer_get(p, out_ext, out_enc)   evaluates p — and everything else recorded so far: who asks for one result will ask for the others —
 *                                        and returns its limbs and / or marshal_binary; either pointer may be NULL
 *   kyb_defer_equal(a, b, eq)            Point::eq;   kyb_defer_flush()   evaluates everything recorded
 *   kyb_defer_stats(out, cap)            nodes recorded, flushes, engine calls made by flushes, Horner chains fused, sums fused,
 *                                        marshal cache hits, nodes in the window now, nodes that left the window, values kept now, values
 *                                        pushed out of the table, answers from the table, operands taken back in (for tests and benchmarks) */
int kyb_defer_input(const int32_t* ext, uint64_t* out);
int kyb_defer_input_enc(const int32_t* ext, const uint8_t* enc, uint64_t* out);
int kyb_defer_null(uint64_t* out);
int kyb_defer_base(uint64_t* out);
int kyb_defer_mul_base(const uint8_t* scalar, uint64_t* out);
int kyb_defer_mul(const uint8_t* scalar, uint64_t p, uint64_t* out);
int kyb_defer_add(uint64_t a, uint64_t b, int subtract, uint64_t* out);
int kyb_defer_neg(uint64_t a, uint64_t* out);
int kyb_defer_get(uint64_t p, int32_t* out_ext, uint8_t* out_enc);
int kyb_defer_equal(uint64_t a, uint64_t b, uint8_t* eq);
int kyb_defer_flush(void);
uint64_t kyb_defer_mark(void);
int kyb_defer_floor(uint64_t mark);
int kyb_defer_stats(uint64_t* out, int cap);

/* ---- PointCanCheckCanonicalAndSmallOrder, group.rs:71-78, point.rs:286-337 ---- */
/* flags[i]: bit 0 = is_canonical(b) — the reference's own expression (point.rs:315-337), which also answers "not canonical" for the 217
 * canonical values y = p-217 .. p-1 (DESIGN.md "Parity definition"); bit 1 = has_small_order().  Exactly one of
 *   enc      n x 32 received bytes.  has_small_order() looks at the canonical re-encoding of the decoded point with the sign bit masked
 *            (point.rs:287-301), i.e. at y mod p, which the bytes give without a square root: no field multiplication at all.  For bytes
 *            that decode to no point the bit is 0 (all five weak y values decode).
 *   pts_ext  n x 40 limbs of points the caller holds: marshalled on the GPU first, as has_small_order(&self) does, then both checks on
 *            those bytes (is_canonical then speaks about the point's own marshal_binary).
 * Callers: schnorr / eddsa verify_with_checks (the batch verification calls do these checks inside); DKG / VSS code checking received
 * points before use. */
int kyb_point_checks_batch(const uint8_t* enc, const int32_t* pts_ext, size_t n, uint8_t* flags);
int kyb_point_checks_batch_dev(const uint8_t* enc, const int32_t* pts_ext, size_t n, uint8_t* flags, void* stream);

/* ---- options ------------------------------------------------------------------------------------ */
/* kyb_set_option / kyb_get_option: deployment knobs of a context — the compute units it works with, the hand-over sizes between the kernel
 * families (all results are the same on either side of one), timing policy (ladder.skip_canonical, mul.short_scalars), the form of out_ext,
 * the host-pointer pipeline, the deferred-point arena, test hooks (diag.*).  KYB_E_BAD_ARG for unknown keys or values.  The library ships
 * ONE kernel per regime: there is no option that selects an algorithm or a kernel variant (see the end of this list).
 *   device.cus        compute units the context's launches are sized for (default 0 = what the device reports).  A host that confines the engine
 *                     to a CU-masked stream, or runs on a partition whose streams see fewer compute units than the device property says, declares
 *                     the number here: the hand-over sizes below — kept as wavefronts per compute unit — and the persistent grids follow it.
 *   coop.max_items    batches of at most this many items take the one-item-per-wavefront kernels (variable base, verification,
 *                     polynomial evaluation; default 24 per compute unit = 6144 on an MI355X, 0 = never); coop.base_max_items the same for the
 *                     fixed base, signing, short sums (18 per compute unit = 4608; a fixed-base multiplication itself — signing: two per signature — leaves them at
 *                     5 per compute unit = 1280 for its mid-size kernel, mul_base.quarters), coop.decode_max_items for a bare decode or encode (4 per compute unit = 1024),
 *                     coop.verify_max_items for the kernels that give ONE item several wavefronts (verification in one launch, signing in one
 *                     launch, the fixed base with four wavefronts per item: item counts up to 2 per compute unit = 512; the variable base with
 *                     an item's scalar in four pieces on four workgroups: up to half of that, 256).  Setting one of them sets an absolute
 *                     item count (until device.cus is set again).  Same results either way.
 *   coop.share_by_load  1 (default): the coop.* and ladder.pair_max_items thresholds are divided by the number of synchronous host-pointer
 *                     calls this process has in flight on the same GPU (each on its own context): kernels that spend 64 or 2 lanes on an item are for a chip
 *                     that would otherwise idle, not for one that 16 threads share.  0: thresholds as set.  Same results either way.
 *   ladder.pair_max_items  ladder launches (variable base, verification, linear combinations) of at most this many items give every item TWO
 *                     lanes of a wavefront, which split the products of a ladder step between them and keep the operand's Montgomery image
 *                     projective, so that no inversion runs in front (default 128 x compute units = one wavefront per SIMD: 32768 on an MI355X, 0 = never): up to there the
 *                     call time is one lane's chain of 255 steps, and the two-lane form takes 0.49 instead of 0.79 ms.  Same results.  These
 *                     launches always walk 256 bits (minus publicly known zeros): ladder.skip_canonical does not apply to them.
 *   ladder.quad_max_items  ... and of at most this many items FOUR lanes, a ladder step three products deep instead of five (variable base from points; default
 *                     64 x compute units = one wavefront per SIMD: 16384, 0 = never): 0.34 instead of 0.43 ms.  Same results.
 *   coop.ladder_max_items  variable base and linear combinations leave the one-item-per-wavefront kernels above this many items (default 14 per compute unit = 3584;
 *                     verification with the keys given as points at 7/8 of it) even when coop.max_items would still allow them: from there the
 *                     two-lane ladder is faster (a plain multiplication of points by full-size scalars already leaves at 9 per compute unit = 2304 while
 *                     ladder.quad_max_items is set: its four-lane ladder is).
 *   coop.ladder_enc_max_items  the same for calls from BYTES — kyb_mul_batch with pts_enc (full-length multipliers), kyb_verify_batch (at 5/6 of it): default 6 per compute unit = 1536;
 *                     above it the role-split launches of ladder.y_only = 2 are faster.
 *   ladder.skip_canonical  1 (default): the batch ladder starts four bits lower when no scalar of the launch reaches 2^252 — true of a scalar
 *                     reduced mod L (L = 2^252 + 2.8e37) except for 2^-127 of them, so the test (an OR over the batch, taken on the way by the
 *                     kernel that prepares the points) says nothing about a canonical secret; one unreduced scalar anywhere and the launch
 *                     walks all 256 bits.  0: always 256.
 *   mul.short_scalars  0 (default): kyb_mul_batch never looks at the size of its scalars.  1: every host-pointer kyb_mul_batch of this context
 *                     behaves like kyb_mul_public_batch (above) — for hosts whose unmodified protocol code cannot say which multipliers are
 *                     public (PubPoly::eval calls Point::mul with x = i + 1): a deployment decision, since a secret below 2^64 (probability
 *                     2^-188 for a uniformly random one) would then be visible in the call's run time.  Same results either way.
 *   ext.projective    0 (default): out_ext always has Z = 1.  1: a small-batch kyb_mul_batch / kyb_mul_base_batch / kyb_pubpoly_eval*_batch /
 *                     kyb_sum_batch / kyb_lincomb_batch call that asks for out_ext ONLY (out_enc == NULL) gets the point as (X : Y : Z : T) with Z != 1 — what the reference's own Point
 *                     holds after a multiplication — and skips the field inversion (one fixed-base call: 55 -> 27 us); every entry
 *                     point accepts such points, kyb_encode_batch pays the inversion when an encoding is wanted
 *   defer.fuse        1 (default): a flush of deferred points (kyb_defer_*) evaluates Horner chains and chains of additions as ONE call each; 0:
 *                     level by level only.  Same results.
 *   defer.max_nodes   the window of the deferred-point arena: the youngest nodes, kept with their graph (default 2^18 = at most 69 MB, at least 16)
 *   defer.keep_mib    MiB of values that evaluated nodes leave behind when the window moves past them (default 256; 0: none — a handle older
 *                     than the window is then refused with KYB_E_STALE)
 *   host.in_place     1 (default): a host-pointer call small enough for host.zero_copy_kib (and of at least 64 KiB) whose array lies in memory
 *                     kyb_host_alloc handed out on this context's device (16-byte aligned start) has its kernels read / write that array where it
 *                     lies instead of a copy in the context's buffer (an 8,192-item multiplication: 0.03-0.05 ms less); other page-locked
 *                     memory may be non-coherent and is copied like pageable memory.  0: always copy.  Same results.
 *   host.zero_copy_kib  host-pointer calls whose arrays together fit this many KiB skip every hipMemcpy: the inputs are copied into the
 *                     context's page-locked buffer by the calling thread and the kernels read and write it over PCIe (default 4096).
 *                     Larger calls of fewer than 2^16 items copy in, run and copy out on the engine stream.
 *   host.pipe_chunks  host-pointer batches of 2^16 items or more are pipelined (one copy-in lane, two compute lanes, one copy-out lane)
 *                     over chunks of 1 1 2 4 4 2 2 .. units; the unit is 1/value of the batch (default 16, 2..64)
 *   host.copy_threads host threads that move pageable batches through the bounce buffers (0 = auto)
 * CROSS-CHECK BUILD ONLY — libkyber_ed25519_hip_crosscheck.so (csrc/Makefile CROSSCHECK=1; test infrastructure, tests/conftest.py `xengine`): the
 * same sources plus the alternative kernels (windowed variable base, radix-16 / -32 fixed base, fused signing, other register budgets and
 * launch shapes) and the selectors below, so that the tests can compare every variant with the product's kernel and with the oracle.  The
 * product library answers KYB_E_BAD_ARG ("unknown option") to each of them and contains none of those kernels.
 *   mul.algo          1 Montgomery ladder + y-recovery (table-free, default), 0 windowed table 1P..8P per lane
 *   mul.ladder_waves  2..4 (default 3): waves per SIMD the ladder kernel's register allocation must allow
 *   mul.select        0 v_cndmask merge, 1 and/or merge of the windowed kernel's table scan
 *   mul.grid_per_cu   1 | 2 workgroups per CU of the windowed kernel (its table workspace is sized for 2)                [mul.algo=0 only]
 *   mul_base.radix    64 (default): 43-window kernel, the table fills a CU's LDS; 32: 52 windows; 16: 64 windows
 *   mul_base.select   0 LDS broadcast scan, 1 ds_bpermute selection                       [radix-16 kernel]
 *   mul_base.block    256 | 512 threads per workgroup                                     [radix-16 kernel]
 *   mul_base.block64  1024 | 512 threads per workgroup of the radix-64 kernel on full batches
 *   mul_base.small_chunks  radix-64 kernel: 256-thread workgroups up to this many chunks per CU (default 2)
 *   finish.batched    1 (default): results stay projective and one inversion serves 8 items (k_finish)
 *   finish.min_items  smallest batch that takes the batched finish / the radix-64, -32 kernels (default 1)
 *   encode.batched    1 (default): kyb_encode_batch shares one inversion between 8 points; 0: one per point
 *   poly.batch_segments  kyb_pubpoly_eval*_batch, long polynomials at 10^3..6x10^4 evaluations: the Horner chain of an evaluation is cut into
 *                     this many segments, one per lane, recombined with x^(s len) mod 8L by the variable-base ladder (0 = chosen by a cost
 *                     model from t, the batch size and the bit length of the largest index; 1 = never; 2..256).  Same results either way.
 *   poly.segments     PubPoly::eval of few evaluations: wavefronts per evaluation (0 = chosen from t and the batch size, 1 = never
 *                     split the Horner chain, 2..32)
 *   ladder.y_only     2 (default): launches of at most ladder.pair_max_items items whose points come as 32-byte ENCODINGS (kyb_mul_batch with
 *                     pts_enc, kyb_verify_batch) run the two-lane ladder on the y of the encoding — u = (1 + y) / (1 - y) needs no x — while
 *                     further workgroups of the SAME launch take the square root of the decode (a verification: of the key and of R, and its hash
 *                     moves into the ladder's workgroups); a short kernel joins them.  1: the decode as a kernel of its own on a side stream.
 *                     0: decode first, then the ladder.  Same results.
 *   mul_base.quarters 1 (default): a fixed-base launch above the one-item-per-wavefront sizes and up to 128 items per compute unit gives every 64 items a
 *                     workgroup of four wavefronts, each adding a quarter of the 43 windows; 0: one lane per item throughout.  Same results.
 *   finish.four       what closes a launch that does not fill the chip (full batches share one field inversion between 8 items of a lane): 2 (default)
 *                     up to 64 x 8 x CUs results (two wavefronts per SIMD) take one inversion per WAVEFRONT, spread over its lanes, Montgomery's trick
 *                     across the 64 lanes; 1 up to 64 x 4 x CUs results take one inversion per 4 items of a lane; 0 per 8 throughout.  Same results.
 *   verify.overlap    1 (default): small verification batches run s*B on a side stream next to the ladder (and, with ladder.y_only, the decode of A)
 *   verify.by_encoding 1 (default): large batches test the equation as enc(s*B - h*A) == R bytes and decode R only on a mismatch
 * Options belong to the calling thread's context. */
int kyb_set_option(const char* key, int value);
int kyb_get_option(const char* key, int* value);
/* per-launch kernel timing for benchmarks: after kyb_profile_begin(m) the next m kernel launches are
 * bracketed by HIP events on their launch stream; kyb_profile_read waits for them and returns
 * (kernel id, milliseconds) pairs in launch order and stops recording.  kyb_profile_begin(0) frees
 * the events.  kyb_kernel_name maps an id to the kernel's name ("k_mul", "k_mul_base", ...). */
int kyb_profile_begin(int max_launches);
int kyb_profile_read(int* kernel_ids, float* ms, int cap, int* count);
const char* kyb_kernel_name(int kernel_id);
/* Benchmark diagnostics (bench.py: the roofline is quoted against a peak and a clock measured in the same run; no product call
 * depends on them).
 *   kyb_diag_mad_peak: every SIMD of the context's GPU issues nothing but v_mad_u64_u32 (8 wavefronts per SIMD, eight independent
 *     accumulator chains each) for at least min_ms milliseconds, once with the unused carry-out written to VCC (as the product kernels do)
 *     and once to an SGPR pair (a few per cent faster in a pure stream); the FASTER run is reported.  *mads_per_s = 32x32+64 multiply-adds per second of the whole chip,
 *     *clock_ghz = the shader clock the chip held under that load (s_memtime / s_memrealtime), *simd_cycles_per_mad = SIMD cycles per
 *     wavefront-wide multiply-add (4.0 = one quarter-rate issue every 4 cycles), *kernel_ms = duration of the measured launch.  Any
 *     pointer may be NULL.  Synchronous.
 *   kyb_diag_wave_stamps(dev_buf): dev_buf = 5 x uint64 of device memory, zeroed by the caller; while set, every wavefront of
 *     k_mul_ladder / k_mul_base64 launched on the context's DEVICE adds (shader cycles, 100 MHz ticks) at its start into words 0, 1
 *     and at its end into words 2, 3, and counts itself in word 4: (w2 - w0) / (w3 - w1) x 100 MHz is the in-kernel clock.  NULL
 *     switches it off (the default: the kernels then execute one extra scalar load and branch at each end).  Waits for the device.
 *     The buffer must stay allocated until the call with NULL — or until the context that set it is released (kyb_ctx_destroy /
 *     kyb_shutdown), which switches the stamps off by itself. */
int kyb_diag_mad_peak(double min_ms, double* mads_per_s, double* clock_ghz, double* simd_cycles_per_mad, double* kernel_ms);
int kyb_diag_wave_stamps(void* dev_buf);
/* kyb_get_option diag.dev_kib / diag.host_kib (read-only): KiB of device / page-locked memory the context's lazily grown buffers hold now. */

/* ---- test hooks: exported by the CROSS-CHECK build only -------------------------------------------------------------------------------
 * libkyber_ed25519_hip_crosscheck.so (csrc/Makefile CROSSCHECK=1, compiled with -DKYB_CROSSCHECK) is the product's sources plus the
 * alternative kernels the tests compare it with, the options that select them, and the hooks below.  The PRODUCT library exports none of
 * them and knows none of these options: nothing in a process that loaded it can make its launches or allocations fail, or read the buffers
 * secrets pass through.  (tests/test_gpu_fault_injection.py, tests/test_gpu_coop.py)
 *   options diag.fail_alloc_after = k / diag.fail_launch_after = k (kyb_set_option; 0 = off): the k-th buffer allocation / kernel launch this
 *     context attempts from now on fails without being made — KYB_E_NOMEM naming the buffer / KYB_E_HIP naming the launch — and the counter
 *     is spent.  A failed call leaves the context usable: the next call gives the ordinary results.
 *   kyb_diag_scratch_read(which, dst, cap, bytes): copies up to cap bytes of one of the context's buffers to dst and stores the buffer's
 *     size in *bytes (0 = not allocated).  which: 0, 1 page-locked zero-copy / bounce buffers; 2 device staging of host-pointer calls;
 *     3 page-locked landing area of large pageable results; 4, 5, 6, 7 the engine stream's scratch (projective records, encodings, products of
 *     small linear combinations, pieces of the four-workgroup variable-base product).  Waits for the device first.
 *   kyb_diag_coop(op, a, b, out): one cooperative one-item-per-wavefront primitive on caller-supplied operands (64 words each).
 *   kyb_diag_phase_stamps(dev_buf): dev_buf = 32 x uint64 of zeroed device memory (NULL = off): the one-item kernels write the constant
 *     100 MHz clock into fixed slots at their phase boundaries (start, scalar multiplication done, partial results combined, inversion
 *     begin / end, stored) — where the time of a one-item call goes (tools/one_item_stamps.py). */
#ifdef KYB_CROSSCHECK
int kyb_diag_scratch_read(int which, uint8_t* dst, size_t cap, size_t* bytes);
int kyb_diag_coop(int op, const uint32_t* a, const uint32_t* b, uint32_t* out);
int kyb_diag_phase_stamps(void* dev_buf);
#endif

#ifdef __cplusplus
}
#endif
#endif /* KYBER_ED25519_H */
