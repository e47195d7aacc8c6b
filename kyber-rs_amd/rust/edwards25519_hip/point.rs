//! `Point` of the HIP group: the type behind `group::edwards25519::Point` when kyber-rs is built with the `hip` feature.
//!
//! ONE code path: every method that does curve arithmetic — `mul`, `add`, `sub`, `neg`, `eq`, `marshal_binary`, `unmarshal_binary`,
//! `has_small_order`, the `pick` / `embed` rejection loop (decode, then `mul` by the cofactor or by the group order) and the batch
//! helpers — reaches the engine (include/kyber_ed25519.h) on the 160-byte `ext` record, which is the reference's own limb layout.
//! The reference's CPU `Point` (`point.rs`) and its `ge.rs` formulas are NOT used: under the feature the reference's `mod point` is
//! compiled out (kyber-rs.hip-feature.patch).  What stays on the host is what involves no field arithmetic: the key-stream handling of
//! `embed`, the length byte of `data`, the byte comparison of `is_canonical`, the formatters (they print the bytes the engine marshals).
//! tools/check_rust_shim.py checks, method by method, that this file reaches the same `kyb_*` entry points as the C++ mirror
//! `host/edwards25519.hpp`, which the GPU tests drive.
//! DEFERRED MODE — the DEFAULT (`set_deferred(false)` per thread or KYBER_HIP_EAGER in the environment switch it off): `mul` / `add` /
//! `sub` / `neg` RECORD their operation in the engine's arena (`kyb_defer_*`) and return a `Point` that holds only a handle; the engine
//! evaluates what was recorded, in batches, when somebody needs bytes or limbs (`marshal_binary`, `eq`, `hash`, `data`, serde, the batch
//! helpers).  Unmodified protocol code then gets one engine call for the t multiplications of `PriPoly::commit` and one for the whole
//! Horner chain of `PubPoly::eval` (tests/cpp/test_vss_round.cpp is the same logic in C++, timed).  Eager mode exists for comparison: an
//! eager `add` of one pair is a round trip to the GPU (27 us; the CPU needs 0.3 us), a recorded one costs a fraction of a microsecond and
//! is evaluated as part of a sum.  `Point` stays `Copy` and `Send`: handles are plain numbers that name the arena (the recording thread's
//! context) they came from, so a point recorded on one thread can be marshalled or multiplied on another.
//! LIFETIME.  A `Copy` point cannot remember what it fetched, and copies of a handle sit in protocol state for the life of a node
//! (dkg.rs:41,170; dss_sig.rs:44), so the ARENA remembers: a node that has been evaluated leaves its value behind when the arena's window
//! (the youngest 2^18 nodes) moves past it, and its handle keeps answering — and keeps working as an operand — from that table (bounded by
//! `defer.keep_mib`, 256 MiB = 1.3 million values; least recently touched first).  Unmodified protocol code needs neither of the two
//! manual controls: `defer_floor(mark)` (end of a round: everything older is dropped, values included — a statement that nothing older is
//! wanted) and `materialize()` (detach one point from the arena for good).  tests/cpp/test_long_running.cpp runs a handle-only client of
//! this shape through the C ABI for hundreds of rounds past the window.
//! Which reference method each one stands for: INTEGRATION.md §3.
use core::fmt::{Debug, Display, Formatter, LowerHex, UpperHex};

use serde::{Deserialize, Serialize};

use crate::{
    cipher::Stream,
    encoding::{BinaryMarshaler, BinaryUnmarshaler, Marshaling, MarshallingError},
    group::{self, edwards25519::Scalar, internal::marshalling, PointCanCheckCanonicalAndSmallOrder, PointError},
};

use std::os::raw::c_int;

use super::ffi::{self, ensure_init, must};

type Limbs = [[i32; 10]; 4];

thread_local! {
    static DEFERRED: std::cell::Cell<bool> = std::cell::Cell::new(std::env::var_os("KYBER_HIP_EAGER").is_none());
}
/// Record this thread's `mul` / `add` / `sub` / `neg` in the engine's arena instead of running them one by one (module docs).
pub fn set_deferred(on: bool) {
    DEFERRED.with(|d| d.set(on));
}
fn deferred() -> bool {
    DEFERRED.with(|d| d.get())
}
/// Everything recorded before `mark` (an earlier `defer_mark()`) may be dropped from the arena: call at the end of a protocol round.
pub fn defer_mark() -> u64 {
    ensure_init();
    unsafe { ffi::kyb_defer_mark() }
}
pub fn defer_floor(mark: u64) {
    ensure_init();
    must(unsafe { ffi::kyb_defer_floor(mark) }, "defer_floor");
}

/// 1 * B as the engine hands it out, asked for once per process
fn base_ext() -> &'static Limbs {
    static BASE_EXT: std::sync::OnceLock<Limbs> = std::sync::OnceLock::new();
    BASE_EXT.get_or_init(|| {
        ensure_init();
        let mut one = [0u8; 32];
        one[0] = 1;
        let mut b: Limbs = [[0; 10]; 4];
        must(unsafe { ffi::kyb_mul_base_batch(one.as_ptr(), 1, std::ptr::null_mut(), b.as_mut_ptr() as *mut i32) }, "base");
        b
    })
}

/// `marshal_id` of the reference's type ("ed.point", point.rs:21)
const MARSHAL_POINT_ID: [u8; 8] = *b"ed.point";
/// the group order L and the cofactor 8 as the 32 little-endian bytes of `PRIME_ORDER_SCALAR` / `COFACTOR_SCALAR` (constants.rs:45-49)
const ORDER_LE: [u8; 32] = [0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10];
const COFACTOR_LE: [u8; 32] = [8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0];
/// marshal_binary of the neutral element
// y of the two classes of order-8 points, little-endian (csrc/consts.inc KYB_W_ORDER8_Y0 / _Y1; constants.rs:3744-3775 lists them among WEAK_KEYS)
const ORDER8_Y0_LE: [u8; 32] = [0x26, 0xe8, 0x95, 0x8f, 0xc2, 0xb2, 0x27, 0xb0, 0x45, 0xc3, 0xf4, 0x89, 0xf2, 0xef, 0x98, 0xf0, 0xd5, 0xdf, 0xac, 0x05, 0xd3, 0xc6, 0x33, 0x39, 0xb1, 0x38, 0x02, 0x88, 0x6d, 0x53, 0xfc, 0x05];
const ORDER8_Y1_LE: [u8; 32] = [0xc7, 0x17, 0x6a, 0x70, 0x3d, 0x4d, 0xd8, 0x4f, 0xba, 0x3c, 0x0b, 0x76, 0x0d, 0x10, 0x67, 0x0f, 0x2a, 0x20, 0x53, 0xfa, 0x2c, 0x39, 0xcc, 0xc6, 0x4e, 0xc7, 0xfd, 0x77, 0x92, 0xac, 0x03, 0x7a];
const NEUTRAL_ENC: [u8; 32] = [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0];

fn invalid_point() -> MarshallingError {
    MarshallingError::InvalidInput("invalid Ed25519 curve point".to_owned())
}

/// X, Y, Z, T as 4 x [i32; 10] radix-2^25.5 limbs — the ABI's `ext` record, which is also what the reference's `Point`
/// keeps in its `ge` field (point.rs:23-27) — plus the reference's `var_time` flag (carried, never acted on: SURVEY.md §2).
/// `pend != 0`: the point has been recorded in the engine's arena and not asked for yet; `ge` is then unset and `limbs()` fetches it.
/// serde sees the two fields of the reference's type (`Plain`): a recorded point is evaluated before it is serialised.
#[derive(Copy, Clone, Debug, Serialize, Deserialize)]
#[serde(from = "Plain", into = "Plain")]
pub struct Point {
    ge: Limbs,
    var_time: bool,
    pend: u64,
    /// The 32 bytes `marshal_binary` returns for this value, when the point was unmarshalled from exactly those bytes.  Protocol code
    /// marshals what it has just unmarshalled (every hash over received commitments and keys), asks `has_small_order` of it
    /// (schnorr_sig.rs:79-95) and compares it (point.rs:227-241 compares ENCODINGS): with the bytes at hand none of these reaches the
    /// engine.  Part of the value: every operation that changes the point drops it.  (`Point` is `Copy`, as in the reference, so a
    /// `&self` marshal cannot store its result; the C++ mirror, host/edwards25519.hpp, also keeps the bytes of a first marshal.)
    enc: Option<[u8; 32]>,
}

#[derive(Serialize, Deserialize)]
#[serde(rename = "Point")]
struct Plain {
    ge: Limbs,
    var_time: bool,
}
impl From<Plain> for Point {
    fn from(p: Plain) -> Self {
        Point { ge: p.ge, var_time: p.var_time, pend: 0, enc: None }
    }
}
impl From<Point> for Plain {
    fn from(p: Point) -> Self {
        Plain { ge: p.limbs(), var_time: p.var_time }
    }
}

impl Default for Point {
    /// all-zero limbs, like the derived default of the reference's element
    fn default() -> Self {
        Point { ge: [[0; 10]; 4], var_time: false, pend: 0, enc: None }
    }
}

// the reference derives these on the limbs; so do we, on the evaluated limbs
impl Eq for Point {}
impl PartialOrd for Point {
    fn partial_cmp(&self, other: &Self) -> Option<core::cmp::Ordering> {
        Some(self.cmp(other))
    }
}
impl Ord for Point {
    fn cmp(&self, other: &Self) -> core::cmp::Ordering {
        (self.limbs(), self.var_time).cmp(&(other.limbs(), other.var_time))
    }
}

impl Point {
    pub fn new() -> Self {
        Self::default()
    }
    fn from_limbs(ge: Limbs) -> Self {
        Point { ge, var_time: false, pend: 0, enc: None }
    }
    /// a point that exists only as a recorded operation so far
    fn recorded(self, handle: u64) -> Self {
        Point { ge: [[0; 10]; 4], pend: handle, enc: None, ..self }
    }
    /// the limbs: held, or evaluated now (with everything else recorded so far) and fetched from the arena
    fn limbs(&self) -> Limbs {
        if self.pend == 0 {
            return self.ge;
        }
        ensure_init();
        let mut out: Limbs = [[0; 10]; 4];
        must(unsafe { ffi::kyb_defer_get(self.pend, out.as_mut_ptr() as *mut i32, std::ptr::null_mut()) }, "defer_get");
        out
    }
    /// this point as the operand of a recorded operation: its handle, or a leaf made of its limbs (the arena recognises limbs it has seen)
    fn handle(&self) -> u64 {
        if self.pend != 0 {
            return self.pend;
        }
        ensure_init();
        let mut h = 0u64;
        // (with the bytes it was unmarshalled from, when it has them: comparing it with an evaluated point is then a byte comparison in the arena)
        let bytes = self.enc.as_ref().map_or(std::ptr::null(), |e| e.as_ptr());
        must(unsafe { ffi::kyb_defer_input_enc(self.ge.as_ptr() as *const i32, bytes, &mut h) }, "defer_input");
        h
    }
    fn ext_mut(&mut self) -> *mut i32 {
        self.pend = 0;
        self.enc = None;
        self.ge.as_mut_ptr() as *mut i32
    }
    /// Detach this point from the arena: its limbs are fetched (evaluating what it depends on) and it no longer names a handle.
    /// Never needed for correctness (module docs, LIFETIME); for a host that calls `defer_floor` and wants one point to survive it.
    pub fn materialize(&mut self) -> Self {
        self.ge = self.limbs();
        self.pend = 0;
        *self
    }

    /// the 32 bytes of `marshal_binary`, from the engine (one field inversion on the GPU; for a recorded point: evaluation in batches,
    /// bytes cached in the arena)
    fn encoding(&self) -> [u8; 32] {
        if let Some(bytes) = self.enc {
            return bytes;
        }
        ensure_init();
        let mut b = [0u8; 32];
        if self.pend != 0 || deferred() {
            // (a point that holds limbs goes through the arena as the leaf its limbs are: the bytes stay with the leaf, so marshalling the
            //  same value again — protocol state is marshalled in every round — costs no engine call although `&self` cannot remember them)
            must(unsafe { ffi::kyb_defer_get(self.handle(), std::ptr::null_mut(), b.as_mut_ptr()) }, "defer_get");
        } else {
            must(unsafe { ffi::kyb_encode_batch(self.ge.as_ptr() as *const i32, 1, b.as_mut_ptr()) }, "encode");
        }
        b
    }
    /// `unmarshal_binary` on the engine: None iff the bytes are not 32 or do not decode
    fn decode(data: &[u8]) -> Option<Limbs> {
        if data.len() != 32 {
            return None;
        }
        ensure_init();
        let (mut ok, mut out) = (0u8, [[0i32; 10]; 4]);
        must(unsafe { ffi::kyb_decode_batch(data.as_ptr(), 1, out.as_mut_ptr() as *mut i32, &mut ok) }, "decode");
        (ok != 0).then_some(out)
    }

    /// lower-case hex of the 32 marshalled bytes (what the reference's formatters print)
    fn hex(&self, upper: bool) -> String {
        self.encoding().iter().map(|byte| if upper { format!("{byte:02X}") } else { format!("{byte:02x}") }).collect()
    }
    /// add / sub of the trait: recorded in deferred mode, else one pair through the engine (`kyb_add_batch`, a batch of one)
    fn add_sub(self, p1: &Self, p2: &Self, subtract: bool) -> Self {
        ensure_init();
        if deferred() {
            let mut h = 0u64;
            must(unsafe { ffi::kyb_defer_add(p1.handle(), p2.handle(), subtract as c_int, &mut h) }, "defer_add");
            return self.recorded(h);
        }
        let (a, b) = (p1.limbs(), p2.limbs());
        let mut out: Limbs = [[0; 10]; 4];
        must(unsafe { ffi::kyb_add_batch(a.as_ptr() as *const i32, b.as_ptr() as *const i32, 1, out.as_mut_ptr() as *mut i32, subtract as c_int) }, "add");
        Point { ge: out, pend: 0, enc: None, ..self }
    }

    /// `mul` for a multiplier the caller KNOWS to be public (a share index, the cofactor): the engine may then skip its leading
    /// zero bits (`kyb_mul_public_batch`; 29 us instead of 158 for a 10-bit index).  Never for a secret.
    pub fn mul_public(mut self, s: &Scalar, p: &Self) -> Self {
        ensure_init();
        let operand = p.limbs();
        must(
            unsafe {
                ffi::kyb_mul_public_batch(s.v.as_ptr(), std::ptr::null(), operand.as_ptr() as *const i32, 1, std::ptr::null_mut(), self.ext_mut(),
                                          std::ptr::null_mut())
            },
            "mul_public",
        );
        self
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
                let inp: Vec<[[i32; 10]; 4]> = ps.iter().map(|q| q.limbs()).collect();
                must(
                    unsafe {
                        ffi::kyb_mul_batch(sc.as_ptr(), std::ptr::null(), inp.as_ptr() as *const i32, n, std::ptr::null_mut(),
                                           staged.as_mut_ptr() as *mut i32, std::ptr::null_mut())
                    },
                    "mul_batch",
                )
            }
        }
        staged.into_iter().map(Point::from_limbs).collect()
    }

    /// 32-byte encodings of many points with one shared inversion per 8 (marshal_binary of each, point.rs:35-41)
    pub fn marshal_batch(ps: &[Point]) -> Vec<[u8; 32]> {
        ensure_init();
        let inp: Vec<[[i32; 10]; 4]> = ps.iter().map(|q| q.limbs()).collect();
        let mut out = vec![[0u8; 32]; ps.len()];
        must(unsafe { ffi::kyb_encode_batch(inp.as_ptr() as *const i32, ps.len(), out.as_mut_ptr() as *mut u8) }, "encode_batch");
        out
    }

    /// out[i] = a[i] + b[i] (or a[i] - b[i]): one engine call for a whole vector (PubPoly::add over t coefficients, poly.rs:486-507)
    pub fn add_batch(a: &[Point], b: &[Point], subtract: bool) -> Vec<Point> {
        ensure_init();
        assert_eq!(a.len(), b.len());
        let ia: Vec<[[i32; 10]; 4]> = a.iter().map(|q| q.limbs()).collect();
        let ib: Vec<[[i32; 10]; 4]> = b.iter().map(|q| q.limbs()).collect();
        let mut staged = vec![[[0i32; 10]; 4]; a.len()];
        must(
            unsafe { ffi::kyb_add_batch(ia.as_ptr() as *const i32, ib.as_ptr() as *const i32, a.len(), staged.as_mut_ptr() as *mut i32, subtract as c_int) },
            "add_batch",
        );
        staged.into_iter().map(Point::from_limbs).collect()
    }

    /// `is_canonical(bytes)` and `has_small_order()` of the decoded point for many RECEIVED encodings in one engine call
    /// (bytes only, no curve arithmetic): what a DKG / VSS node asks of every point it is sent (point.rs:286-337)
    pub fn checks_batch(encs: &[[u8; 32]]) -> Vec<(bool, bool)> {
        ensure_init();
        let mut flags = vec![0u8; encs.len()];
        must(unsafe { ffi::kyb_point_checks_batch(encs.as_ptr() as *const u8, std::ptr::null(), encs.len(), flags.as_mut_ptr()) }, "point_checks_batch");
        flags.into_iter().map(|f| (f & 1 != 0, f & 2 != 0)).collect()
    }

    /// a[i] == b[i] for many pairs without any inversion (Point::eq pays two per pair, point.rs:227-241)
    pub fn eq_batch(a: &[Point], b: &[Point]) -> Vec<bool> {
        ensure_init();
        assert_eq!(a.len(), b.len());
        let ia: Vec<[[i32; 10]; 4]> = a.iter().map(|q| q.limbs()).collect();
        let ib: Vec<[[i32; 10]; 4]> = b.iter().map(|q| q.limbs()).collect();
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
    let inp: Vec<[[i32; 10]; 4]> = shares.iter().map(|q| q.limbs()).collect();
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
        return Err(invalid_point());
    }
    Ok(staged.into_iter().map(Point::from_limbs).collect())
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
        return Err(invalid_point());
    }
    Ok(staged.into_iter().map(Point::from_limbs).collect())
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

/// Are these 32 bytes, which decode, exactly what `marshal_binary` yields for the point they decode to?  Not when y >= p (`fe_from_bytes`
/// accepts it, `fe_to_bytes` reduces it: fe.rs:52-122) and not when x = 0 carries a sign bit (ge.rs:124-179 accepts it, the encoder writes 0).
fn is_the_canonical_encoding(b: &[u8]) -> bool {
    let (mut ones, mut any) = (0xffu8, 0u8);
    for byte in &b[1..31] {
        ones &= byte;
        any |= byte;
    }
    let top = b[31] & 0x7f;
    let y_ge_p = ones == 0xff && top == 0x7f && b[0] >= 0xed;
    let y_is_one = any == 0 && top == 0 && b[0] == 1;
    let y_is_minus_one = ones == 0xff && top == 0x7f && b[0] == 0xec;
    !y_ge_p && !(b[31] & 0x80 != 0 && (y_is_one || y_is_minus_one))
}

/// The comparison of point.rs:286-313 on bytes that ARE the point's encoding: sign bit masked, against y = 0, 1, p - 1 and the two classes of
/// order-8 points (the WEAK_KEYS below p, constants.rs:3744-3775; csrc/verify.h `pt_has_small_order` is the engine's form of it).
fn small_order_encoding(e: &[u8; 32]) -> bool {
    let mut y = *e;
    y[31] &= 0x7f;
    let mut one = [0u8; 32];
    one[0] = 1;
    let mut minus_one = [0xffu8; 32];
    minus_one[0] = 0xec;
    minus_one[31] = 0x7f;
    let matches = |w: &[u8; 32]| y.iter().zip(w).fold(0u8, |d, (a, b)| d | (a ^ b)) == 0;
    matches(&[0u8; 32]) | matches(&one) | matches(&minus_one) | matches(&ORDER8_Y0_LE) | matches(&ORDER8_Y1_LE)
}

impl BinaryMarshaler for Point {
    fn marshal_binary(&self) -> Result<Vec<u8>, MarshallingError> {
        Ok(self.encoding().to_vec())
    }
}
impl BinaryUnmarshaler for Point {
    fn unmarshal_binary(&mut self, data: &[u8]) -> Result<(), MarshallingError> {
        self.ge = Self::decode(data).ok_or_else(invalid_point)?;
        self.pend = 0;
        self.enc = is_the_canonical_encoding(data).then(|| data.try_into().unwrap());
        Ok(())
    }
}

impl Marshaling for Point {
    // the crate's generic point (un)marshalling helpers, as every group's Point uses them
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

    /// the neutral element (0 : 1 : 1 : 0)
    fn null(self) -> Self {
        let mut ge: Limbs = [[0; 10]; 4];
        ge[1][0] = 1;
        ge[2][0] = 1;
        Point { ge, pend: 0, enc: None, ..self }
    }

    /// 1 * B from the engine (the reference copies a literal; the same point)
    fn base(self) -> Self {
        Point { ge: *base_ext(), pend: 0, enc: None, ..self }
    }

    /// `embed` without data (point.rs:90-92)
    fn pick<S: Stream>(self, rand: &mut S) -> Self {
        self.embed(None, rand)
    }

    /// point.rs:94-97 copies the element; here the value is limbs, handle AND the bytes it was unmarshalled from (`var_time` stays the
    /// receiver's, as in the reference, whose `set` assigns `ge` only)
    fn set(&mut self, p: &Self) -> Self {
        *self = Point { var_time: self.var_time, ..*p };
        *self
    }

    /// 29: a length byte below and a byte of randomness above the data (point.rs:99-104)
    fn embed_len(&self) -> usize {
        (255 - 8 - 8) / 8
    }

    /// The rejection loop of point.rs:106-167 with its curve arithmetic on the engine: a candidate is 32 bytes of key stream (the
    /// length and the data laid over bytes 0..=dl); `kyb_decode_batch` says whether it is a point; without data the candidate times
    /// the cofactor is the result unless that is the neutral element; with data the candidate itself is, provided it lies in the
    /// prime-order subgroup (candidate times L is the neutral element).  Eager calls in either mode: every answer decides the next step.
    fn embed<S: Stream>(self, data: Option<&[u8]>, rand: &mut S) -> Self {
        ensure_init();
        let dl = data.map_or(0, |d| d.len().min(self.embed_len()));
        loop {
            let mut cand = [0u8; 32];
            rand.xor_key_stream(&mut cand, &[0u8; 32]).unwrap();
            if let Some(d) = data {
                cand[0] = dl as u8;
                cand[1..=dl].copy_from_slice(&d[..dl]);
            }
            let Some(limbs) = Self::decode(&cand) else { continue };
            let (scalar, keep_product) = if data.is_none() { (&COFACTOR_LE, true) } else { (&ORDER_LE, false) };
            let (mut enc, mut product) = ([0u8; 32], [[0i32; 10]; 4]);
            must(
                unsafe {
                    ffi::kyb_mul_batch(scalar.as_ptr(), std::ptr::null(), limbs.as_ptr() as *const i32, 1, enc.as_mut_ptr(), product.as_mut_ptr() as *mut i32,
                                       std::ptr::null_mut())
                },
                "embed",
            );
            if keep_product && enc != NEUTRAL_ENC {
                return Point { ge: product, pend: 0, enc: None, ..self };
            }
            if !keep_product && enc == NEUTRAL_ENC {
                return Point { ge: limbs, pend: 0, enc: None, ..self };
            }
        }
    }

    /// the bytes `embed` placed behind the length byte of the encoding
    fn data(&self) -> Result<Vec<u8>, PointError> {
        let bytes = self.encoding();
        let len = usize::from(bytes[0]);
        if len > self.embed_len() {
            return Err(PointError::EmbedDataLength);
        }
        Ok(bytes[1..=len].to_vec())
    }

    /// One pair through the engine; deferred mode: recorded (a chain of additions is evaluated as one sum).  Vectors: `Point::add_batch`.
    fn add(self, p1: &Self, p2: &Self) -> Self {
        self.add_sub(p1, p2, false)
    }

    fn sub(self, p1: &Self, p2: &Self) -> Self {
        self.add_sub(p1, p2, true)
    }

    /// -(X : Y : Z : T) = (-X : Y : Z : -T), limb by limb; no engine call (deferred mode: recorded)
    fn neg(&mut self, a: &Self) -> Self {
        if deferred() {
            ensure_init();
            let mut h = 0u64;
            must(unsafe { ffi::kyb_defer_neg(a.handle(), &mut h) }, "defer_neg");
            *self = self.recorded(h);
            return *self;
        }
        let flip = |f: &[i32; 10]| f.map(|limb| -limb);
        let l = a.limbs();
        self.ge = [flip(&l[0]), l[1], l[2], flip(&l[3])];
        self.pend = 0;
        self.enc = None;
        *self
    }

    /// None -> fixed base, Some(P) -> variable base; the scalar is used as stored (`s.v`, no reduction).
    /// Every in-tree caller passes the generator as Some(base) (PriPoly::commit; vss): when the operand is, limb for limb, what
    /// `base()` hands out and the scalar is below 2^255 (no top-digit quirk in either routine), the fixed-base kernel gives the
    /// same point in a sixth of the time.
    fn mul(mut self, s: &Scalar, p: Option<&Self>) -> Self {
        ensure_init();
        let fixed = match p {
            None => true,
            Some(q) => q.pend == 0 && q.ge == *base_ext() && s.v[31] & 0x80 == 0,
        };
        if deferred() {
            let mut h = 0u64;
            let rc = if fixed {
                unsafe { ffi::kyb_defer_mul_base(s.v.as_ptr(), &mut h) }
            } else {
                unsafe { ffi::kyb_defer_mul(s.v.as_ptr(), p.unwrap().handle(), &mut h) }
            };
            must(rc, "defer_mul");
            return self.recorded(h);
        }
        let rc = if fixed {
            unsafe { ffi::kyb_mul_base_batch(s.v.as_ptr(), 1, std::ptr::null_mut(), self.ext_mut()) }
        } else {
            let operand = p.unwrap().limbs();
            unsafe { ffi::kyb_mul_batch(s.v.as_ptr(), std::ptr::null(), operand.as_ptr() as *const i32, 1, std::ptr::null_mut(), self.ext_mut(), std::ptr::null_mut()) }
        };
        must(rc, "mul");
        self
    }
}

impl PartialEq for Point {
    /// the reference compares the two encodings (two inversions); the engine compares projectively, same answer
    fn eq(&self, p2: &Self) -> bool {
        if let (Some(a), Some(b)) = (self.enc, p2.enc) {
            return a == b; // what the reference compares
        }
        ensure_init();
        let mut e = 0u8;
        if self.pend != 0 || p2.pend != 0 {
            must(unsafe { ffi::kyb_defer_equal(self.handle(), p2.handle(), &mut e) }, "defer_equal");
        } else {
            must(unsafe { ffi::kyb_equal_batch(self.ge.as_ptr() as *const i32, p2.ge.as_ptr() as *const i32, 1, &mut e) }, "eq");
        }
        e != 0
    }
}

impl core::hash::Hash for Point {
    fn hash<H: std::hash::Hasher>(&self, state: &mut H) {
        self.encoding().hash(state);
    }
}

// formatting: the reference's three formatters print the marshalled bytes; so do these
impl Display for Point {
    fn fmt(&self, f: &mut Formatter<'_>) -> core::fmt::Result {
        write!(f, "Ed25519Point(0x{})", self.hex(false))
    }
}
impl LowerHex for Point {
    fn fmt(&self, f: &mut Formatter<'_>) -> core::fmt::Result {
        f.write_str(if f.alternate() { "0x" } else { "" })?;
        f.write_str(&self.hex(false))
    }
}
impl UpperHex for Point {
    fn fmt(&self, f: &mut Formatter<'_>) -> core::fmt::Result {
        f.write_str(if f.alternate() { "0X" } else { "" })?;
        f.write_str(&self.hex(true))
    }
}

// the two checks of verify_with_checks (`verify_batch` runs them on the GPU for whole batches, `Point::checks_batch` for received bytes)
impl PointCanCheckCanonicalAndSmallOrder for Point {
    /// on the engine: marshals the limbs and compares with the weak keys (no `unmarshal`, hence total on any limbs)
    fn has_small_order(&self) -> bool {
        if let Some(bytes) = self.enc {
            return small_order_encoding(&bytes);
        }
        ensure_init();
        let mut flags = 0u8;
        let l = self.limbs();
        must(unsafe { ffi::kyb_point_checks_batch(std::ptr::null(), l.as_ptr() as *const i32, 1, &mut flags) }, "point_checks");
        flags & 2 != 0
    }
    /// Byte logic on the caller's buffer, no decoding: false for the encodings point.rs:315-337 refuses — bytes 1..=30 all 0xff, the
    /// low seven bits of byte 31 all set, and byte 0 at least 0x14.  (The reference computes 0xED - (1 - b0) in wrapping 16-bit
    /// arithmetic and looks at bit 8, which is set from b0 = 0x14 on — libsodium's form would say 0xED; csrc/verify.h has the derivation
    /// and the GPU's verify kernels use the same threshold.)  Every byte is looked at whatever the others hold.
    fn is_canonical(&self, b: &[u8]) -> bool {
        if b.len() != 32 {
            return false;
        }
        let mut high_differs = (b[31] & 0x7f) ^ 0x7f;
        for byte in &b[1..31] {
            high_differs |= byte ^ 0xff;
        }
        let high_all_ones = high_differs == 0;
        let low_too_big = b[0] >= 0x14;
        !(high_all_ones & low_too_big)
    }
}
