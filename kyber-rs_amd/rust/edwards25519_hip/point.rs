//! `Point` of the HIP group: every method of `impl group::Point for Point` in
//! src/group/edwards25519/point.rs:75-225 (and the marshaling / comparison / formatting impls around it), with the
//! curve arithmetic forwarded to the engine as batch-of-1 calls.  Host-side logic (the `embed` rejection loop, `data`,
//! `has_small_order`, `is_canonical`) is the reference's own, expression for expression.
use core::fmt::{Debug, Display, Formatter, LowerHex, UpperHex};

use serde::{Deserialize, Serialize};

use crate::{
    cipher::Stream,
    encoding::{BinaryMarshaler, BinaryUnmarshaler, Marshaling, MarshallingError},
    group::{
        self,
        edwards25519::constants::{COFACTOR_SCALAR, PRIME_ORDER_SCALAR, WEAK_KEYS},
        edwards25519::ge::{CachedGroupElement, CompletedGroupElement, ExtendedGroupElement},
        edwards25519::Scalar,
        internal::marshalling,
        PointCanCheckCanonicalAndSmallOrder, PointError,
    },
};

use std::os::raw::c_int;

use super::ffi::{self, ensure_init, must};

/// 1 * B from the engine, asked for once per process
fn base_ext() -> &'static [[i32; 10]; 4] {
    static BASE_EXT: std::sync::OnceLock<[[i32; 10]; 4]> = std::sync::OnceLock::new();
    BASE_EXT.get_or_init(|| {
        ensure_init();
        let mut one = [0u8; 32];
        one[0] = 1;
        let mut b = [[0i32; 10]; 4];
        must(unsafe { ffi::kyb_mul_base_batch(one.as_ptr(), 1, std::ptr::null_mut(), b.as_mut_ptr() as *mut i32) }, "base");
        b
    })
}

const MARSHAL_POINT_ID: [u8; 8] = [b'e', b'd', b'.', b'p', b'o', b'i', b'n', b't'];

/// Same data as the reference's `Point { ge: ExtendedGroupElement, var_time: bool }` (point.rs:23-27):
/// `ge` = X, Y, Z, T as 4 x [i32; 10] radix-2^25.5 limbs — the layout of the ABI's `ext` records.
#[derive(Copy, Clone, Eq, Ord, PartialOrd, Debug, Serialize, Deserialize)]
pub struct Point {
    ge: [[i32; 10]; 4],
    var_time: bool,
}

impl Default for Point {
    /// `ExtendedGroupElement::default()` is all-zero limbs (ge.rs:78-83 derives Default); kept as is
    fn default() -> Self {
        Point { ge: [[0; 10]; 4], var_time: false }
    }
}

impl Point {
    pub fn new() -> Self {
        Self::default()
    }
    /// the same limbs as the reference's element: the ABI's `ext` record IS X, Y, Z, T in fe_from_bytes' normal form (or the
    /// tight limbs of a product), so the CPU formulas of ge.rs accept it as it is
    fn to_ref(&self) -> ExtendedGroupElement {
        ExtendedGroupElement { x: self.ge[0], y: self.ge[1], z: self.ge[2], t: self.ge[3] }
    }
    fn ext(&self) -> *const i32 {
        self.ge.as_ptr() as *const i32
    }
    fn ext_mut(&mut self) -> *mut i32 {
        self.ge.as_mut_ptr() as *mut i32
    }
    /// `ExtendedGroupElement::write_bytes` (ge.rs:112-122)
    fn write_bytes(&self, b: &mut [u8; 32]) {
        ensure_init();
        must(unsafe { ffi::kyb_encode_batch(self.ext(), 1, b.as_mut_ptr()) }, "encode");
    }
    /// `ExtendedGroupElement::set_bytes` (ge.rs:124-179): false iff the length is not 32 or no square root exists
    fn set_bytes(&mut self, data: &[u8]) -> bool {
        if data.len() != 32 {
            return false;
        }
        ensure_init();
        let mut ok = 0u8;
        let mut out = [[0i32; 10]; 4];
        must(unsafe { ffi::kyb_decode_batch(data.as_ptr(), 1, out.as_mut_ptr() as *mut i32, &mut ok) }, "decode");
        if ok == 0 {
            return false;
        }
        self.ge = out;
        true
    }

    // ---- throughput entry points for callers that own a batch (PriPoly::commit, poly.rs:195-206) ----

    /// out[i] = s[i] * p[i]   (or s[i] * B when `p` is None): ONE engine call instead of n trait calls
    pub fn mul_batch(s: &[Scalar], p: Option<&[Point]>) -> Vec<Point> {
        ensure_init();
        let n = s.len();
        let sc: Vec<u8> = s.iter().flat_map(|x| x.v).collect();
        let mut staged = vec![[[0i32; 10]; 4]; n]; // 160-byte stride (Point itself carries `var_time` behind the limbs)
        match p {
            None => must(unsafe { ffi::kyb_mul_base_batch(sc.as_ptr(), n, std::ptr::null_mut(), staged.as_mut_ptr() as *mut i32) }, "mul_base_batch"),
            Some(ps) => {
                assert_eq!(ps.len(), n);
                let inp: Vec<[[i32; 10]; 4]> = ps.iter().map(|q| q.ge).collect();
                must(
                    unsafe {
                        ffi::kyb_mul_batch(sc.as_ptr(), std::ptr::null(), inp.as_ptr() as *const i32, n, std::ptr::null_mut(),
                                           staged.as_mut_ptr() as *mut i32, std::ptr::null_mut())
                    },
                    "mul_batch",
                )
            }
        }
        staged.into_iter().map(|ge| Point { ge, var_time: false }).collect()
    }

    /// 32-byte encodings of many points with one shared inversion per 8 (marshal_binary of each, point.rs:35-41)
    pub fn marshal_batch(ps: &[Point]) -> Vec<[u8; 32]> {
        ensure_init();
        let inp: Vec<[[i32; 10]; 4]> = ps.iter().map(|q| q.ge).collect();
        let mut out = vec![[0u8; 32]; ps.len()];
        must(unsafe { ffi::kyb_encode_batch(inp.as_ptr() as *const i32, ps.len(), out.as_mut_ptr() as *mut u8) }, "encode_batch");
        out
    }

    /// out[i] = a[i] + b[i] (or a[i] - b[i]): one engine call for a whole vector (PubPoly::add over t coefficients, poly.rs:486-507)
    pub fn add_batch(a: &[Point], b: &[Point], subtract: bool) -> Vec<Point> {
        ensure_init();
        assert_eq!(a.len(), b.len());
        let ia: Vec<[[i32; 10]; 4]> = a.iter().map(|q| q.ge).collect();
        let ib: Vec<[[i32; 10]; 4]> = b.iter().map(|q| q.ge).collect();
        let mut staged = vec![[[0i32; 10]; 4]; a.len()];
        must(
            unsafe { ffi::kyb_add_batch(ia.as_ptr() as *const i32, ib.as_ptr() as *const i32, a.len(), staged.as_mut_ptr() as *mut i32, subtract as c_int) },
            "add_batch",
        );
        staged.into_iter().map(|ge| Point { ge, var_time: false }).collect()
    }

    /// a[i] == b[i] for many pairs without any inversion (Point::eq pays two per pair, point.rs:227-241)
    pub fn eq_batch(a: &[Point], b: &[Point]) -> Vec<bool> {
        ensure_init();
        assert_eq!(a.len(), b.len());
        let ia: Vec<[[i32; 10]; 4]> = a.iter().map(|q| q.ge).collect();
        let ib: Vec<[[i32; 10]; 4]> = b.iter().map(|q| q.ge).collect();
        let mut eq = vec![0u8; a.len()];
        must(unsafe { ffi::kyb_equal_batch(ia.as_ptr() as *const i32, ib.as_ptr() as *const i32, a.len(), eq.as_mut_ptr()) }, "equal_batch");
        eq.into_iter().map(|e| e != 0).collect()
    }
}

/// `recover_commit` (poly.rs:566-603) with the accumulation on the GPU: the Lagrange coefficients stay scalar
/// arithmetic on the CPU, the t multiplications and additions are one `kyb_lincomb_batch` call.
pub fn recover_commit_accumulate(lagrange: &[Scalar], shares: &[Point]) -> Point {
    ensure_init();
    assert_eq!(lagrange.len(), shares.len());
    let sc: Vec<u8> = lagrange.iter().flat_map(|x| x.v).collect();
    let inp: Vec<[[i32; 10]; 4]> = shares.iter().map(|q| q.ge).collect();
    let mut out = Point::default();
    must(
        unsafe {
            ffi::kyb_lincomb_batch(sc.as_ptr(), std::ptr::null(), inp.as_ptr() as *const i32, 0, 1, shares.len(), std::ptr::null_mut(),
                                   out.ext_mut(), std::ptr::null_mut())
        },
        "lincomb",
    );
    out
}

/// The verifier's side of a DKG round on the deals as they arrive: `commits_enc` holds the `t` 32-byte commitments of every
/// dealer, dealer after dealer (`Deal.commitments`, vss/pedersen/vss.rs:113-124, before any `unmarshal_binary`), `idx[g]` is the
/// index dealer g's polynomial is evaluated at (vss.rs:904-909: the verifier's own).  One engine call decodes the m·t points
/// on the GPU and returns the m evaluations; an encoding that is not a point is the reference's unmarshal error.
pub fn eval_each_wire(commits_enc: &[u8], t: usize, idx: &[u32]) -> Result<Vec<Point>, MarshallingError> {
    ensure_init();
    let m = idx.len();
    assert!(t > 0 && commits_enc.len() == 32 * t * m);
    let mut staged = vec![[[0i32; 10]; 4]; m];
    let mut ok = vec![0u8; m * t];
    must(
        unsafe {
            ffi::kyb_pubpoly_eval_multi_enc_batch(commits_enc.as_ptr(), t, m, idx.as_ptr(), 1, std::ptr::null_mut(), staged.as_mut_ptr() as *mut i32,
                                                  ok.as_mut_ptr())
        },
        "pubpoly_eval_multi_enc",
    );
    if ok.iter().any(|&f| f == 0) {
        return Err(MarshallingError::InvalidInput("invalid Ed25519 curve point".to_owned()));
    }
    Ok(staged.into_iter().map(|ge| Point { ge, var_time: false }).collect())
}

/// The distributed public polynomial (dkg.rs:905-953 folds `PubPoly::add`, poly.rs:486-507, over the dealers) from the same
/// buffer: coefficient j of the result = sum over the dealers of their commitment j.
pub fn sum_polys_wire(commits_enc: &[u8], t: usize) -> Result<Vec<Point>, MarshallingError> {
    ensure_init();
    assert!(t > 0 && commits_enc.len() % (32 * t) == 0 && !commits_enc.is_empty());
    let dealers = commits_enc.len() / (32 * t);
    let mut staged = vec![[[0i32; 10]; 4]; t];
    let mut ok = vec![0u8; dealers * t];
    must(
        unsafe { ffi::kyb_sum_enc_batch(commits_enc.as_ptr(), t, dealers, 1, std::ptr::null_mut(), staged.as_mut_ptr() as *mut i32, ok.as_mut_ptr()) },
        "sum_enc",
    );
    if ok.iter().any(|&f| f == 0) {
        return Err(MarshallingError::InvalidInput("invalid Ed25519 curve point".to_owned()));
    }
    Ok(staged.into_iter().map(|ge| Point { ge, var_time: false }).collect())
}

/// `schnorr::verify_with_checks` / `eddsa::verify_with_checks` for a batch (every DKG deal / response / DSS partial
/// signature is verified by every peer).  Returns the per-item status: 0 = valid, else the reference's FIRST failing
/// check in the order of the chosen flavour (codes in include/kyber_ed25519.h).
pub fn verify_batch(pubs: &[[u8; 32]], msgs: &[&[u8]], sigs: &[[u8; 64]], eddsa_order: bool) -> Vec<u8> {
    ensure_init();
    let n = pubs.len();
    assert!(msgs.len() == n && sigs.len() == n);
    let mut off = Vec::with_capacity(n + 1);
    let mut blob = Vec::new();
    off.push(0u32);
    for m in msgs {
        blob.extend_from_slice(m);
        off.push(u32::try_from(blob.len()).expect("message blob of 4 GiB or more"));
    }
    blob.push(0);
    let mut status = vec![0u8; n];
    must(
        unsafe {
            ffi::kyb_verify_batch(pubs.as_ptr() as *const u8, blob.as_ptr(), off.as_ptr(), sigs.as_ptr() as *const u8, n,
                                  if eddsa_order { 0 } else { 1 }, status.as_mut_ptr())
        },
        "verify_batch",
    );
    status
}

impl BinaryMarshaler for Point {
    // point.rs:35-41
    fn marshal_binary(&self) -> Result<Vec<u8>, MarshallingError> {
        let mut b = [0_u8; 32];
        self.write_bytes(&mut b);
        Ok(b.to_vec())
    }
}
impl BinaryUnmarshaler for Point {
    // point.rs:43-50
    fn unmarshal_binary(&mut self, data: &[u8]) -> Result<(), MarshallingError> {
        if !self.set_bytes(data) {
            return Err(MarshallingError::InvalidInput("invalid Ed25519 curve point".to_owned()));
        }
        Ok(())
    }
}

impl Marshaling for Point {
    // point.rs:53-73
    fn marshal_to(&self, w: &mut impl std::io::Write) -> Result<(), MarshallingError> {
        marshalling::point_marshal_to(self, w)
    }
    fn marshal_size(&self) -> usize {
        32
    }
    fn unmarshal_from(&mut self, r: &mut impl std::io::Read) -> Result<(), MarshallingError> {
        marshalling::point_unmarshal_from(self, r)
    }
    fn unmarshal_from_random(&mut self, r: &mut (impl std::io::Read + Stream)) {
        marshalling::point_unmarshal_from_random(self, r);
    }
    fn marshal_id(&self) -> [u8; 8] {
        MARSHAL_POINT_ID
    }
}

impl group::Point for Point {
    type SCALAR = Scalar;

    /// point.rs:79-82 — `ge.zero()`: (0 : 1 : 1 : 0)
    fn null(mut self) -> Self {
        self.ge = [[0; 10]; 4];
        self.ge[1][0] = 1;
        self.ge[2][0] = 1;
        self
    }

    /// point.rs:85-88 — the reference copies the literal BASEEXT; here 1 * B from the engine, once per process (the same point; like BASEEXT not normalised to Z = 1 when ext.projective is on)
    fn base(mut self) -> Self {
        self.ge = *base_ext();
        self
    }

    /// point.rs:90-92
    fn pick<S: Stream>(self, rand: &mut S) -> Self {
        self.embed(None, rand)
    }

    /// point.rs:94-97
    fn set(&mut self, p: &Self) -> Self {
        self.ge = p.ge;
        *self
    }

    /// point.rs:99-104
    fn embed_len(&self) -> usize {
        (255 - 8 - 8) / 8
    }

    /// point.rs:106-167 — the rejection loop is host logic; decode, the cofactor / order multiplications and the
    /// comparisons with the neutral element go to the engine.
    fn embed<S: Stream>(mut self, data: Option<&[u8]>, rand: &mut S) -> Self {
        let mut dl = self.embed_len();
        let data_len = match data {
            Some(d) => d.len(),
            None => 0,
        };
        if dl > data_len {
            dl = data_len;
        }
        let null_point = Point::default().null();
        loop {
            let mut b = [0_u8; 32];
            rand.xor_key_stream(&mut b, &[0_u8; 32]).unwrap();
            if let Some(d) = data {
                b[0] = dl as u8;
                b[1..1 + dl].copy_from_slice(&d[0..dl]);
            }
            if !self.set_bytes(&b) {
                continue;
            }
            if data.is_none() {
                let old_self = &self.clone();
                self = self.mul(&COFACTOR_SCALAR, Some(old_self));
                if self.eq(&null_point) {
                    continue;
                }
                return self;
            }
            let mut q = Point::default();
            q = q.mul(&PRIME_ORDER_SCALAR, Some(&self));
            if q.eq(&null_point) {
                return self;
            }
        }
    }

    /// point.rs:169-177
    fn data(&self) -> Result<Vec<u8>, PointError> {
        let mut b = [0u8; 32];
        self.write_bytes(&mut b);
        let dl = b[0] as usize;
        if dl > self.embed_len() {
            return Err(PointError::EmbedDataLength);
        }
        Ok(b[1..1 + dl].to_vec())
    }

    /// point.rs:179-188.  A single pair is added on the CPU with the reference's own formulas (ge.rs:99-110, 217-234, 292-297): nine field
    /// multiplications take 0.3 us there, a batch-of-1 round trip to the GPU 26 us, and the ABI's ext record IS the reference's limb
    /// layout, so nothing is converted.  Cargo feature `hip-single-add` sends it to the engine instead (`kyb_add_batch`, n = 1: the same
    /// point); vectors go there in any case (`Point::add_batch`).
    fn add(mut self, p1: &Self, p2: &Self) -> Self {
        if cfg!(feature = "hip-single-add") {
            ensure_init();
            must(unsafe { ffi::kyb_add_batch(p1.ext(), p2.ext(), 1, self.ext_mut(), 0) }, "add");
            return self;
        }
        let mut t2 = CachedGroupElement::default();
        let mut r = CompletedGroupElement::default();
        p2.to_ref().write_cached(&mut t2);
        r.add(&p1.to_ref(), &t2);
        let mut out = ExtendedGroupElement::default();
        r.to_extended(&mut out);
        self.ge = [out.x, out.y, out.z, out.t];
        self
    }

    /// point.rs:190-199, as `add`
    fn sub(mut self, p1: &Self, p2: &Self) -> Self {
        if cfg!(feature = "hip-single-add") {
            ensure_init();
            must(unsafe { ffi::kyb_add_batch(p1.ext(), p2.ext(), 1, self.ext_mut(), 1) }, "sub");
            return self;
        }
        let mut t2 = CachedGroupElement::default();
        let mut r = CompletedGroupElement::default();
        p2.to_ref().write_cached(&mut t2);
        r.sub(&p1.to_ref(), &t2);
        let mut out = ExtendedGroupElement::default();
        r.to_extended(&mut out);
        self.ge = [out.x, out.y, out.z, out.t];
        self
    }

    /// point.rs:201-204 — `ge.neg`: X and T negated limb by limb (ge.rs:86-91), no engine call
    fn neg(&mut self, a: &Self) -> Self {
        for i in 0..10 {
            self.ge[0][i] = -a.ge[0][i];
            self.ge[1][i] = a.ge[1][i];
            self.ge[2][i] = a.ge[2][i];
            self.ge[3][i] = -a.ge[3][i];
        }
        *self
    }

    /// point.rs:207-224 — None -> fixed base (ge_scalar_mult_base), Some(P) -> variable base (ge_scalar_mult; the
    /// reference's var_time branch is unreachable).  The scalar is used as stored (`s.v`, no reduction).
    fn mul(mut self, s: &Scalar, p: Option<&Self>) -> Self {
        ensure_init();
        match p {
            None => must(unsafe { ffi::kyb_mul_base_batch(s.v.as_ptr(), 1, std::ptr::null_mut(), self.ext_mut()) }, "mul(None)"),
            // Every in-tree caller passes the generator as Some(base) (PriPoly::commit, poly.rs:195-206; vss.rs:303): when the operand is,
            // limb for limb, what `base()` hands out and the scalar is below 2^255 (no top-digit quirk in either routine), the fixed-base
            // kernel gives the same point in a sixth of the time.
            Some(a_p) if a_p.ge == *base_ext() && s.v[31] & 0x80 == 0 => {
                must(unsafe { ffi::kyb_mul_base_batch(s.v.as_ptr(), 1, std::ptr::null_mut(), self.ext_mut()) }, "mul(Some(base))")
            }
            Some(a_p) => must(
                unsafe {
                    ffi::kyb_mul_batch(s.v.as_ptr(), std::ptr::null(), a_p.ext(), 1, std::ptr::null_mut(), self.ext_mut(), std::ptr::null_mut())
                },
                "mul(Some)",
            ),
        }
        self
    }
}

impl PartialEq for Point {
    /// point.rs:227-241 compares the two encodings; the engine compares projectively (no inversion), same answer
    fn eq(&self, p2: &Self) -> bool {
        ensure_init();
        let mut e = 0u8;
        must(unsafe { ffi::kyb_equal_batch(self.ext(), p2.ext(), 1, &mut e) }, "eq");
        e != 0
    }
}

impl core::hash::Hash for Point {
    // point.rs:243-249
    fn hash<H: std::hash::Hasher>(&self, state: &mut H) {
        let mut b = [0_u8; 32];
        self.write_bytes(&mut b);
        b.hash(state);
    }
}

impl Display for Point {
    fn fmt(&self, f: &mut Formatter<'_>) -> core::fmt::Result {
        write!(f, "Ed25519Point({self:#x})")
    }
}
impl LowerHex for Point {
    fn fmt(&self, f: &mut Formatter<'_>) -> core::fmt::Result {
        let prefix = if f.alternate() { "0x" } else { "" };
        let mut b = [0u8; 32];
        self.write_bytes(&mut b);
        write!(f, "{prefix}{}", hex::encode(b))
    }
}
impl UpperHex for Point {
    fn fmt(&self, f: &mut Formatter<'_>) -> core::fmt::Result {
        let prefix = if f.alternate() { "0X" } else { "" };
        let mut b = [0u8; 32];
        self.write_bytes(&mut b);
        write!(f, "{prefix}{}", hex::encode_upper(b))
    }
}

impl PointCanCheckCanonicalAndSmallOrder for Point {
    /// point.rs:286-313
    fn has_small_order(&self) -> bool {
        let s = match self.marshal_binary() {
            Ok(v) => v,
            Err(_) => return false,
        };
        let mut c = [0u8; 5];
        (0..31).for_each(|j| {
            for i in 0..5 {
                c[i] |= s[j] ^ WEAK_KEYS[i][j];
            }
        });
        for i in 0..5 {
            c[i] |= (s[31] & 0x7f) ^ WEAK_KEYS[i][31];
        }
        let mut k = 0;
        (0..5).for_each(|i| {
            k |= (c[i] as u16).wrapping_sub(1);
        });
        (k >> 8) & 1 > 0
    }

    /// point.rs:315-337, expression for expression (including its `0xED - (1 - b0)`; csrc/verify.h has the analysis)
    fn is_canonical(&self, b: &[u8]) -> bool {
        if b.len() != 32 {
            return false;
        }
        let mut c = (b[31] & 0x7f) ^ 0x7f;
        for i in (1..=30).rev() {
            c |= b[i] ^ 0xff;
        }
        c = ((c as u16).wrapping_sub(1) >> 8) as u8;
        let d = ((0xEDu16.wrapping_sub(1u16.wrapping_sub(b[0] as u16))) >> 8) as u8;
        1 - (c & d & 1) == 1
    }
}
