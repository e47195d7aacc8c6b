//! Rust shim: `impl kyber_rs::group::Point` forwarding to the MI355X engine's C ABI
//! (include/kyber_ed25519.h).  UNBUILT in this repository (no cargo/rustc in the image).
//!
//! It replaces `kyber_rs::group::edwards25519::Point` (src/group/edwards25519/point.rs:23-241):
//! same in-memory representation (`ExtendedGroupElement` = 4 x [i32; 10] + `var_time`), same trait
//! methods, so DKG / VSS / DSS code generic over `Group` runs unmodified.  Every curve operation is
//! a batch-of-1 call; callers that own a batch (PriPoly::commit, poly.rs:195-206) use `mul_batch`.
//! `mul` & co. are infallible in the trait (group.rs:139): an engine failure panics.
use std::os::raw::{c_char, c_int, c_void};

#[allow(non_camel_case_types)]
type size_t = usize;

extern "C" {
    pub fn kyb_abi_version() -> c_int; // == 1 for this file
    pub fn kyb_init(device: c_int) -> c_int;
    pub fn kyb_shutdown();
    pub fn kyb_last_error() -> *const c_char;
    pub fn kyb_mul_base_batch(scalars: *const u8, n: size_t, out_enc: *mut u8, out_ext: *mut i32) -> c_int;
    pub fn kyb_mul_batch(scalars: *const u8, pts_enc: *const u8, pts_ext: *const i32, n: size_t,
                         out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_add_batch(a_ext: *const i32, b_ext: *const i32, n: size_t, out_ext: *mut i32, subtract: c_int) -> c_int;
    pub fn kyb_encode_batch(pts_ext: *const i32, n: size_t, out_enc: *mut u8) -> c_int;
    pub fn kyb_decode_batch(enc: *const u8, n: size_t, out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_schnorr_sign_batch(x: *const u8, k: *const u8, msgs: *const u8, msg_off: *const u32,
                                  n: size_t, sig: *mut u8) -> c_int;
    pub fn kyb_mul_batch_dev(scalars: *const u8, pts_enc: *const u8, pts_ext: *const i32, n: size_t,
                             out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8, stream: *mut c_void) -> c_int;
    // callers that hold the public key (EdDSA objects, DSS long-term keys): one fixed-base mult per signature
    pub fn kyb_schnorr_sign_keyed_batch(x: *const u8, pubs: *const u8, k: *const u8, msgs: *const u8, msg_off: *const u32,
                                        n: size_t, sig: *mut u8) -> c_int;
    pub fn kyb_eddsa_sign_batch(seeds: *const u8, msgs: *const u8, msg_off: *const u32, n: size_t, sig: *mut u8, pub_out: *mut u8) -> c_int;
    pub fn kyb_eddsa_sign_keyed_batch(seeds: *const u8, pubs: *const u8, msgs: *const u8, msg_off: *const u32, n: size_t, sig: *mut u8) -> c_int;
    // verify_with_checks (eddsa_sig.rs:159-212 = flavor 0, schnorr_sig.rs:53-110 = flavor 1): status[i] 0 = valid, else the first failing check
    pub fn kyb_verify_batch(pubs: *const u8, msgs: *const u8, msg_off: *const u32, sigs: *const u8, n: size_t, flavor: c_int, status: *mut u8) -> c_int;
    // PubPoly::eval / shares (poly.rs:457-478), recover_commit / recover_pub_poly accumulation (poly.rs:566-634), Point::eq
    pub fn kyb_pubpoly_eval_batch(commits_ext: *const i32, t: size_t, indices: *const u32, n: size_t, out_enc: *mut u8, out_ext: *mut i32) -> c_int;
    pub fn kyb_lincomb_batch(scalars: *const u8, pts_enc: *const u8, pts_ext: *const i32, shared_points: c_int, m: size_t, t: size_t,
                             out_enc: *mut u8, out_ext: *mut i32, ok: *mut u8) -> c_int;
    pub fn kyb_equal_batch(a_ext: *const i32, b_ext: *const i32, n: size_t, eq: *mut u8) -> c_int;
    // page-locked batch buffers (optional: pageable slices work, through the engine's bounce buffers)
    pub fn kyb_host_alloc(bytes: size_t) -> *mut c_void;
    pub fn kyb_host_free(p: *mut c_void);
}

fn must(rc: c_int, what: &str) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(kyb_last_error()) }.to_string_lossy().into_owned();
        panic!("kyber-ed25519-hip: {what} failed ({rc}): {msg}");
    }
}

/// Same layout as the reference's `Point { ge: ExtendedGroupElement, var_time: bool }` (point.rs:23-27).
#[derive(Copy, Clone, Debug, Default, serde::Serialize, serde::Deserialize)]
#[repr(C)]
pub struct Point {
    pub ge: [[i32; 10]; 4], // X, Y, Z, T
    var_time: bool,
}

pub type Scalar = kyber_rs::group::edwards25519::Scalar; // unchanged: stays on the CPU (scalar.rs)

impl Point {
    fn ext(&self) -> *const i32 { self.ge.as_ptr() as *const i32 }
    fn ext_mut(&mut self) -> *mut i32 { self.ge.as_mut_ptr() as *mut i32 }

    /// throughput entry point: out[i] = s[i] * p[i] (or s[i] * B when `p` is None)
    pub fn mul_batch(s: &[Scalar], p: Option<&[Point]>) -> Vec<Point> {
        let n = s.len();
        let sc: Vec<u8> = s.iter().flat_map(|x| x.v).collect();
        let mut out = vec![Point::default(); n];
        let mut staged = vec![[[0i32; 10]; 4]; n];   // 160-byte stride (Point itself carries `var_time` behind the limbs)
        match p {
            None => must(unsafe { kyb_mul_base_batch(sc.as_ptr(), n, std::ptr::null_mut(), staged.as_mut_ptr() as *mut i32) }, "mul_base_batch"),
            Some(ps) => {
                let inp: Vec<[[i32; 10]; 4]> = ps.iter().map(|q| q.ge).collect();
                must(unsafe { kyb_mul_batch(sc.as_ptr(), std::ptr::null(), inp.as_ptr() as *const i32, n,
                                            std::ptr::null_mut(), staged.as_mut_ptr() as *mut i32, std::ptr::null_mut()) }, "mul_batch")
            }
        }
        for (o, g) in out.iter_mut().zip(staged) { o.ge = g; }
        out
    }
}

/// `recover_commit` (poly.rs:566-603) with the accumulation on the GPU: the Lagrange coefficients stay scalar
/// arithmetic on the CPU, the t multiplications and additions are one `kyb_lincomb_batch` call.
pub fn recover_commit_accumulate(lagrange: &[Scalar], shares: &[Point]) -> Point {
    assert_eq!(lagrange.len(), shares.len());
    let sc: Vec<u8> = lagrange.iter().flat_map(|x| x.v).collect();
    let inp: Vec<[[i32; 10]; 4]> = shares.iter().map(|q| q.ge).collect();
    let mut out = Point::default();
    must(unsafe { kyb_lincomb_batch(sc.as_ptr(), std::ptr::null(), inp.as_ptr() as *const i32, 0, 1, shares.len(),
                                    std::ptr::null_mut(), out.ext_mut(), std::ptr::null_mut()) }, "lincomb");
    out
}

/// `schnorr::verify_with_checks` for a batch (every DKG deal / response / DSS partial signature is verified by every peer).
/// msgs[i] are concatenated; returns the per-item status (0 = valid; see include/kyber_ed25519.h for the codes).
pub fn verify_batch(pubs: &[[u8; 32]], msgs: &[&[u8]], sigs: &[[u8; 64]], eddsa_order: bool) -> Vec<u8> {
    let n = pubs.len();
    let mut off = Vec::with_capacity(n + 1);
    let mut blob = Vec::new();
    off.push(0u32);
    for m in msgs { blob.extend_from_slice(m); off.push(blob.len() as u32); }
    blob.push(0);
    let mut status = vec![0u8; n];
    must(unsafe { kyb_verify_batch(pubs.as_ptr() as *const u8, blob.as_ptr(), off.as_ptr(), sigs.as_ptr() as *const u8, n,
                                   if eddsa_order { 0 } else { 1 }, status.as_mut_ptr()) }, "verify_batch");
    status
}

impl kyber_rs::encoding::BinaryMarshaler for Point {
    fn marshal_binary(&self) -> Result<Vec<u8>, kyber_rs::encoding::MarshallingError> {          // point.rs:35-41
        let mut b = vec![0u8; 32];
        must(unsafe { kyb_encode_batch(self.ext(), 1, b.as_mut_ptr()) }, "encode");
        Ok(b)
    }
}
impl kyber_rs::encoding::BinaryUnmarshaler for Point {
    fn unmarshal_binary(&mut self, data: &[u8]) -> Result<(), kyber_rs::encoding::MarshallingError> { // point.rs:43-50
        let mut ok = 0u8;
        let mut out = [[0i32; 10]; 4];
        if data.len() == 32 {
            must(unsafe { kyb_decode_batch(data.as_ptr(), 1, out.as_mut_ptr() as *mut i32, &mut ok) }, "decode");
        }
        if data.len() != 32 || ok == 0 {
            return Err(kyber_rs::encoding::MarshallingError::InvalidInput("invalid Ed25519 curve point".to_owned()));
        }
        self.ge = out;
        Ok(())
    }
}

impl PartialEq for Point {                                                                       // point.rs:227-241
    fn eq(&self, o: &Self) -> bool {
        use kyber_rs::encoding::BinaryMarshaler;
        self.marshal_binary().unwrap() == o.marshal_binary().unwrap()
    }
}

/// The `group::Point` methods (group.rs:85-140).  `pick` / `embed` / `data` / `has_small_order` /
/// `is_canonical` keep the reference's host logic (point.rs:90-177, 286-337) and reach the engine only
/// through `unmarshal_binary`, `mul` and `marshal_binary` above; they are omitted here for brevity.
impl Point {
    pub fn null(mut self) -> Self { self.ge = [[0; 10]; 4]; self.ge[1][0] = 1; self.ge[2][0] = 1; self }   // point.rs:79-82
    pub fn base(mut self) -> Self {                                                                        // point.rs:85-88
        let mut one = [0u8; 32]; one[0] = 1;
        must(unsafe { kyb_mul_base_batch(one.as_ptr(), 1, std::ptr::null_mut(), self.ext_mut()) }, "base");
        self
    }
    pub fn add(mut self, a: &Self, b: &Self) -> Self {                                                     // point.rs:179-188
        must(unsafe { kyb_add_batch(a.ext(), b.ext(), 1, self.ext_mut(), 0) }, "add"); self
    }
    pub fn sub(mut self, a: &Self, b: &Self) -> Self {                                                     // point.rs:190-199
        must(unsafe { kyb_add_batch(a.ext(), b.ext(), 1, self.ext_mut(), 1) }, "sub"); self
    }
    pub fn neg(&mut self, a: &Self) -> Self {                                                              // point.rs:201-204
        for i in 0..10 { self.ge[0][i] = -a.ge[0][i]; self.ge[1][i] = a.ge[1][i]; self.ge[2][i] = a.ge[2][i]; self.ge[3][i] = -a.ge[3][i]; }
        *self
    }
    pub fn mul(mut self, s: &Scalar, p: Option<&Self>) -> Self {                                           // point.rs:207-224
        match p {
            None => must(unsafe { kyb_mul_base_batch(s.v.as_ptr(), 1, std::ptr::null_mut(), self.ext_mut()) }, "mul(None)"),
            Some(q) => must(unsafe { kyb_mul_batch(s.v.as_ptr(), std::ptr::null(), q.ext(), 1, std::ptr::null_mut(), self.ext_mut(), std::ptr::null_mut()) }, "mul(Some)"),
        }
        self
    }
}
