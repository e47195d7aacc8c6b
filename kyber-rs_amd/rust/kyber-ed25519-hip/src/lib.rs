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
        let out_ext = out.as_mut_ptr() as *mut i32; // NB: stride must be 160 B — use a [[i32;10];4] staging Vec in real code
        let _ = out_ext;
        let mut staged = vec![[[0i32; 10]; 4]; n];
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
