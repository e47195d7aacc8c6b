//! `group::edwards25519_hip` — the `Point` of kyber-rs's Ed25519 group with its curve arithmetic on an MI355X.
//!
//! UNBUILT SOURCE (this repository's image has no Rust toolchain; parity evidence comes from the same C ABI driven by the C++
//! mirror `host/*.hpp` and the Python tests).  It is an IN-CRATE module of kyber-rs, enabled by a cargo feature:
//!
//!   1. copy this directory to `src/group/edwards25519_hip/`;
//!   2. apply `../kyber-rs.hip-feature.patch` (shown in INTEGRATION.md §3) to the reference (ten lines: `pub mod edwards25519_hip;` under the feature, the
//!      re-export `group::edwards25519::Point` switched by the feature, two modules made `pub(crate)`);
//!   3. (the same patch adds the two `cargo:` link lines to `build.rs`: `libkyber_ed25519_hip.so` is built by
//!      `python __graft_entry__.py build`).
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
//! microseconds on the CPU).  This module holds FFI forwarding only: host-side logic (the `embed` rejection loop, `data`,
//! `has_small_order`, `is_canonical`, the formatters) is delegated to the reference's CPU `Point`.
pub mod ffi;

mod point;

pub use point::{eval_each_wire, recover_commit_accumulate, sum_polys_wire, verify_batch, Point};
