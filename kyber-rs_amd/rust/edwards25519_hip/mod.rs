//! `group::edwards25519_hip` — the Ed25519 group of kyber-rs with its curve arithmetic on an MI355X.
//!
//! UNBUILT SOURCE (this repository's image has no Rust toolchain; parity evidence comes from the same C ABI driven
//! by the C++ mirror `host/*.hpp` and the Python tests).  It is written as an IN-CRATE module of kyber-rs:
//!
//!   1. copy this directory to `src/group/edwards25519_hip/`,
//!   2. add `pub mod edwards25519_hip;` next to `pub mod edwards25519;` in `src/group/mod.rs`,
//!   3. add the two `cargo:` lines of `build.rs.snippet` to the crate's `build.rs`
//!      (links `libkyber_ed25519_hip.so`, built by `python __graft_entry__.py build`),
//!   4. make the CPU group's element arithmetic visible inside the crate: `mod ge;` -> `pub(crate) mod ge;` in
//!      `src/group/edwards25519/mod.rs` (`Point::add` / `sub` of a single pair stay on the CPU, with the reference's own
//!      formulas: nine field multiplications are not worth a round trip to the GPU).
//!
//! In-crate because the reference keeps what a drop-in needs behind crate-private paths
//! (`group::edwards25519::constants` is `pub(crate)`, `group::internal::marshalling` is reached through
//! `crate::group::internal`).  Everything above the group — `share::poly`, `share::vss`, `share::dkg`,
//! `sign::{schnorr, eddsa, dss}`, `dh`, `util::key` — is generic over `Group` / `Suite` and runs unmodified on
//! `SuiteEd25519Hip`:
//!
//! ```ignore
//! let suite = SuiteEd25519Hip::new_blake3_sha256_ed25519();      // kyb_init(0) on first use
//! let dkg = dkg::new_dist_key_generator(suite, &longterm, &participants, t)?;   // share/dkg/pedersen/dkg.rs
//! ```
//!
//! `Scalar` is the reference's own type (mod-L arithmetic costs microseconds and stays on the CPU).
//! Method-for-method correspondence with `src/group/edwards25519/point.rs:75-225`: INTEGRATION.md §3.
pub mod ffi;

mod curve;
mod point;
mod suite;

pub use curve::CurveHip;
pub use point::{eval_each_wire, recover_commit_accumulate, sum_polys_wire, verify_batch, Point};
pub use suite::SuiteEd25519Hip;

/// the scalar type is shared with the CPU group
pub use crate::group::edwards25519::Scalar;
