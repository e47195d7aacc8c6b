//! `extern "C"` block for include/kyber_ed25519.h (ABI version 2).  One line per entry point the Rust side binds.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};
use std::sync::OnceLock;

pub type size_t = usize;
/// opaque `kyb_ctx` / `kyb_group`
#[repr(C)] pub struct kyb_ctx { _p: [u8; 0] }
#[repr(C)] pub struct kyb_group { _p: [u8; 0] }

pub const KYB_ABI_VERSION: c_int = 2;

extern "C" {
    pub fn kyb_abi_version() -> c_int;
    pub fn kyb_set_option(key: *const c_char, value: c_int) -> c_int;
    pub fn kyb_last_error() -> *const c_char;
    pub fn kyb_sync(stream: *mut c_void) -> c_int;
    pub fn kyb_stream_release(stream: *mut c_void) -> c_int;
    // contexts and multi-device groups (one Rust process driving every GPU of a node)
    pub fn kyb_ctx_create(device: c_int, build_table: c_int, out: *mut *mut kyb_ctx) -> c_int;
    pub fn kyb_ctx_destroy(ctx: *mut kyb_ctx) -> c_int;
    pub fn kyb_ctx_set_current(ctx: *mut kyb_ctx) -> c_int;
    pub fn kyb_base_table_export(dst_host: *mut u8) -> c_int;
    pub fn kyb_base_table_import(src_host: *const u8) -> c_int;
    pub fn kyb_group_create(devices: *const c_int, n: c_int, out: *mut *mut kyb_group) -> c_int;
    pub fn kyb_group_destroy(g: *mut kyb_group);
    pub fn kyb_group_size(g: *const kyb_group) -> c_int;
    pub fn kyb_group_ctx(g: *mut kyb_group, rank: c_int) -> *mut kyb_ctx;
    pub fn kyb_group_table_transport(g: *const kyb_group) -> *const c_char;
    pub fn kyb_group_mul_base_batch(g: *mut kyb_group, scalars: *const u8, n: size_t, out_enc: *mut u8, out_ext: *mut i32) -> c_int;
    pub fn kyb_group_mul_batch(g: *mut kyb_group, scalars: *const u8, pts_enc: *const u8, pts_ext: *const i32, n: size_t,
                               out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_group_schnorr_sign_batch(g: *mut kyb_group, x: *const u8, k: *const u8, msgs: *const u8, msg_off: *const u32, n: size_t, sig: *mut u8) -> c_int;
    pub fn kyb_group_verify_batch(g: *mut kyb_group, pubs: *const u8, msgs: *const u8, msg_off: *const u32, sigs: *const u8, n: size_t,
                                  flavor: c_int, status: *mut u8) -> c_int;
    // Point::mul / add / sub / marshal / unmarshal / eq
    pub fn kyb_mul_base_batch(scalars: *const u8, n: size_t, out_enc: *mut u8, out_ext: *mut i32) -> c_int;
    pub fn kyb_mul_batch(scalars: *const u8, pts_enc: *const u8, pts_ext: *const i32, n: size_t,
                         out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_mul_public_batch(scalars: *const u8, pts_enc: *const u8, pts_ext: *const i32, n: size_t,
                                out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_mul_batch_dev(scalars: *const u8, pts_enc: *const u8, pts_ext: *const i32, n: size_t,
                             out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8, stream: *mut c_void) -> c_int;
    pub fn kyb_add_batch(a_ext: *const i32, b_ext: *const i32, n: size_t, out_ext: *mut i32, subtract: c_int) -> c_int;
    pub fn kyb_encode_batch(pts_ext: *const i32, n: size_t, out_enc: *mut u8) -> c_int;
    pub fn kyb_decode_batch(enc: *const u8, n: size_t, out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_equal_batch(a_ext: *const i32, b_ext: *const i32, n: size_t, eq: *mut u8) -> c_int;
    pub fn kyb_point_checks_batch(enc: *const u8, pts_ext: *const i32, n: size_t, flags: *mut u8) -> c_int;
    // signing / verification
    pub fn kyb_schnorr_sign_batch(x: *const u8, k: *const u8, msgs: *const u8, msg_off: *const u32, n: size_t, sig: *mut u8) -> c_int;
    pub fn kyb_schnorr_sign_keyed_batch(x: *const u8, pubs: *const u8, k: *const u8, msgs: *const u8, msg_off: *const u32, n: size_t, sig: *mut u8) -> c_int;
    pub fn kyb_eddsa_sign_batch(seeds: *const u8, msgs: *const u8, msg_off: *const u32, n: size_t, sig: *mut u8, pub_out: *mut u8) -> c_int;
    pub fn kyb_eddsa_sign_keyed_batch(seeds: *const u8, pubs: *const u8, msgs: *const u8, msg_off: *const u32, n: size_t, sig: *mut u8) -> c_int;
    pub fn kyb_verify_batch(pubs: *const u8, msgs: *const u8, msg_off: *const u32, sigs: *const u8, n: size_t, flavor: c_int, status: *mut u8) -> c_int;
    // public polynomials
    pub fn kyb_pubpoly_eval_batch(commits_ext: *const i32, t: size_t, indices: *const u32, n: size_t, out_enc: *mut u8, out_ext: *mut i32) -> c_int;
    pub fn kyb_pubpoly_eval_multi_batch(commits_ext: *const i32, t: size_t, m: size_t, indices: *const u32, k: size_t, out_enc: *mut u8, out_ext: *mut i32) -> c_int;
    pub fn kyb_lincomb_batch(scalars: *const u8, pts_enc: *const u8, pts_ext: *const i32, shared_points: c_int, m: size_t, t: size_t,
                             out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_sum_batch(pts_ext: *const i32, m: size_t, t: size_t, out_enc: *mut u8, out_ext: *mut i32) -> c_int;
    // the same two on wire encodings (Deal.commitments as received)
    pub fn kyb_pubpoly_eval_multi_enc_batch(commits_enc: *const u8, t: size_t, m: size_t, indices: *const u32, k: size_t, out_enc: *mut u8,
                                            out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_sum_enc_batch(pts_enc: *const u8, m: size_t, t: size_t, item_major: c_int, out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8) -> c_int;
    // page-locked batch buffers (optional: pageable slices work, through the engine's bounce buffers)
    // round 3: a verifier's DKG round in one call, public-multiplier combinations, Lagrange coefficients, a dealer's private shares
    pub fn kyb_dkg_verify_round_enc(commits_enc: *const u8, t: size_t, m: size_t, index: u32, eval_enc: *mut u8, eval_ext: *mut i32,
                                    sums_enc: *mut u8, sums_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_lincomb_public_batch(scalars: *const u8, pts_enc: *const u8, pts_ext: *const i32, shared_points: c_int, m: size_t, t: size_t,
                                    out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_lagrange_coeffs_batch(indices: *const u32, m: size_t, t: size_t, out_scalars: *mut u8) -> c_int;
    pub fn kyb_verify_points_batch(pubs_ext: *const i32, msgs: *const u8, msg_off: *const u32, sigs: *const u8, n: size_t, flavor: c_int, status: *mut u8) -> c_int;
    pub fn kyb_pripoly_eval_batch(coeffs: *const u8, m: size_t, t: size_t, indices: *const u32, k: size_t, out_shares: *mut u8) -> c_int;
    // deferred points: operations recorded in the context's arena, evaluated in batches when a result is asked for
    pub fn kyb_defer_input(ext: *const i32, out: *mut u64) -> c_int;
    pub fn kyb_defer_input_enc(ext: *const i32, enc: *const u8, out: *mut u64) -> c_int;
    pub fn kyb_defer_mul_base(scalar: *const u8, out: *mut u64) -> c_int;
    pub fn kyb_defer_mul(scalar: *const u8, p: u64, out: *mut u64) -> c_int;
    pub fn kyb_defer_add(a: u64, b: u64, subtract: c_int, out: *mut u64) -> c_int;
    pub fn kyb_defer_neg(a: u64, out: *mut u64) -> c_int;
    pub fn kyb_defer_get(p: u64, out_ext: *mut i32, out_enc: *mut u8) -> c_int;
    pub fn kyb_defer_equal(a: u64, b: u64, eq: *mut u8) -> c_int;
    pub fn kyb_defer_flush() -> c_int;
    pub fn kyb_defer_mark() -> u64;
    pub fn kyb_defer_floor(mark: u64) -> c_int;
    pub fn kyb_host_alloc(bytes: size_t) -> *mut c_void;
    pub fn kyb_host_free(p: *mut c_void);
}

pub const KYB_BASE_TABLE_BYTES: usize = 335232;

fn device() -> c_int {
    std::env::var("KYBER_HIP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0)
}

/// The base-point table image, built ONCE per process (on the GPU, by a short-lived context) and kept on the host: every thread's
/// context imports it (a 327 KiB copy and a checksum) instead of rebuilding it.
fn table_image() -> &'static [u8] {
    static IMAGE: OnceLock<Vec<u8>> = OnceLock::new();
    IMAGE.get_or_init(|| {
        unsafe {
            assert_eq!(kyb_abi_version(), KYB_ABI_VERSION, "libkyber_ed25519_hip.so has another ABI version");
        }
        let mut c: *mut kyb_ctx = std::ptr::null_mut();
        must(unsafe { kyb_ctx_create(device(), 1, &mut c) }, "kyb_ctx_create");
        must(unsafe { kyb_ctx_set_current(c) }, "kyb_ctx_set_current");
        let mut img = vec![0u8; KYB_BASE_TABLE_BYTES];
        must(unsafe { kyb_base_table_export(img.as_mut_ptr()) }, "kyb_base_table_export");
        unsafe {
            kyb_ctx_set_current(std::ptr::null_mut());
            kyb_ctx_destroy(c);
        }
        img
    })
}

/// Host-pointer calls of ONE context are serialised (its staging buffers and streams are one set), so every thread that touches a
/// `Point` gets a context of its own on first use: N threads then run N small calls on the GPU at the same time
/// (`profiles/r02/concurrent_small_calls.log`: 16 threads, 175,000 `mul(s, None)` per second; start the process with
/// GPU_MAX_HW_QUEUES=16 for that — the library leaves the environment alone).  Options belong to a context.
struct ThreadCtx(*mut kyb_ctx);
impl ThreadCtx {
    fn new() -> Self {
        let img = table_image();
        let mut c: *mut kyb_ctx = std::ptr::null_mut();
        must(unsafe { kyb_ctx_create(device(), 0, &mut c) }, "kyb_ctx_create");
        must(unsafe { kyb_ctx_set_current(c) }, "kyb_ctx_set_current");
        must(unsafe { kyb_base_table_import(img.as_ptr()) }, "kyb_base_table_import");
        // a Point keeps extended coordinates, as the reference's does: take them projective (no inversion per multiplication;
        // marshal_binary pays it when an encoding is wanted)
        must(unsafe { kyb_set_option(b"ext.projective\0".as_ptr() as *const c_char, 1) }, "kyb_set_option(ext.projective)");
        // A deployment that accepts it (include/kyber_ed25519.h, mul.short_scalars) lets every small host-pointer mul whose scalars are all
        // below 2^64 take a short ladder: the Horner step `mul(x_i, Some(v))` of an unmodified PubPoly::eval then costs 29 us, not 158.
        if std::env::var_os("KYBER_HIP_SHORT_PUBLIC_SCALARS").is_some() {
            must(unsafe { kyb_set_option(b"mul.short_scalars\0".as_ptr() as *const c_char, 1) }, "kyb_set_option(mul.short_scalars)");
        }
        ThreadCtx(c)
    }
}
impl Drop for ThreadCtx {
    /// Runs at thread exit — for the main thread possibly while the process is already tearing the HIP runtime down: both calls
    /// return an error code in that case and nothing here looks at it.
    fn drop(&mut self) {
        unsafe {
            let _ = kyb_ctx_set_current(std::ptr::null_mut());
            let _ = kyb_ctx_destroy(self.0);
        }
    }
}
thread_local! {
    static TL_CTX: ThreadCtx = ThreadCtx::new();
}

/// A context for the calling thread on first use (table image built once per process, imported per thread); every method that
/// reaches the engine calls it first (a thread-local access afterwards).  There is no process-wide default context.
pub fn ensure_init() {
    TL_CTX.with(|_| ());
}

/// The trait methods are infallible (`mul` returns `Self`, group.rs:139): an engine failure cannot be reported and panics.
pub fn must(rc: c_int, what: &str) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(kyb_last_error()) }.to_string_lossy().into_owned();
        panic!("kyber edwards25519_hip: {what} failed ({rc}): {msg}");
    }
}
