//! `group::edwards25519_hip` — the `Point` of kyber-rs's Ed25519 group with its curve arithmetic on an MI355X.
//!
//! UNBUILT SOURCE (this repository's image has no Rust toolchain; parity evidence comes from the same C ABI driven by the C++
//! mirror `host/*.hpp` and the Python tests, and `tools/check_rust_shim.py` ties every method of this module to the `kyb_*` entry
//! points the mirror's method of the same name reaches).  It is an IN-CRATE module of kyber-rs, enabled by a cargo feature:
//!
//!   1. copy this directory to `src/group/edwards25519_hip/`;
//!   2. apply `../kyber-rs.hip-feature.patch` (a `diff -u` against the reference tree, shown in INTEGRATION.md §3): the feature `hip` in
//!      `Cargo.toml`, a new `build.rs` with the two `cargo:` link lines, `pub mod edwards25519_hip;` in `src/group.rs` under the feature,
//!      and in `src/group/edwards25519/mod.rs` the re-export `Point` switched by the feature (the reference's `mod point` is compiled
//!      out under it: it compares itself with `constants::NULL_POINT`, whose type is the re-exported name);
//!   3. build with `KYBER_ED25519_HIP_LIB_DIR=<dir of libkyber_ed25519_hip.so> cargo build --features hip`
//!      (`python __graft_entry__.py build` makes the library).
//!
//! With `--features hip` the NAME `group::edwards25519::Point` resolves to the type of this module, so the reference's own
//! `Curve` and `SuiteEd25519` — which are written against that name — hand out engine-backed points without being touched,
//! copied or wrapped, and everything above the group (`share::poly`, `share::vss`, `share::dkg`, `sign::{schnorr, eddsa, dss}`,
//! `dh`, `util::key`), generic over `Group` / `Suite`, runs unmodified:
//!
//! ```ignore
//! let suite = SuiteEd25519::new_blake3_sha256_ed25519();        // the reference's suite; Point::mul now runs on the GPU
//! let dkg = dkg::new_dist_key_generator(suite, &longterm, &participants, t)?;
//! ```
//!
//! Without the feature the crate is byte for byte what it was.  `Scalar` stays the reference's type (mod-L arithmetic costs
//! microseconds on the CPU).  ONE code path: every curve operation of `Point` — single additions and the `pick` / `embed` rejection
//! loop included — reaches the engine; the reference's CPU point and its `ge.rs` formulas are not used (point.rs, module docs).
pub mod ffi;

mod point;

pub use point::{eval_each_wire, recover_commit_accumulate, sum_polys_wire, verify_batch, Point};
