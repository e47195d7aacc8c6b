//! `CurveHip` — `impl Group` whose `Point` is the engine-backed one (mirror of src/group/edwards25519/curve.rs:19-125).
use core::fmt::{Debug, Display, Formatter};

use serde::{Deserialize, Serialize};
use sha2::{Digest, Sha256, Sha512};

use crate::dh::Dh;
use crate::group::edwards25519::{CurveError, Scalar};
use crate::group::Group;
use crate::util::key::{Generator, KeyError};
use crate::util::random;

use super::Point;

#[derive(Clone, Copy, Debug, Serialize, Deserialize)]
pub struct CurveHip {}

impl Dh for CurveHip {
    type H = Sha256; // curve.rs:24-26; dh_exchange's default body is suite.point().mul(..) -> the engine
}

impl Display for CurveHip {
    fn fmt(&self, f: &mut Formatter<'_>) -> core::fmt::Result {
        write!(f, "Ed25519") // curve.rs:28-32: the same group, the same name
    }
}

impl Group for CurveHip {
    type POINT = Point;

    fn scalar(&self) -> Scalar {
        Scalar::new() // curve.rs:42-44
    }
    fn scalar_len(&self) -> usize {
        32 // curve.rs:48-50
    }
    fn point(&self) -> Point {
        Point::new() // curve.rs:52-54
    }
    fn point_len(&self) -> usize {
        32 // curve.rs:57-59
    }
    fn is_prime_order(&self) -> Option<bool> {
        None // curve.rs:61-63
    }
}

impl CurveHip {
    pub const fn new() -> Self {
        CurveHip {}
    }

    /// curve.rs:74-87: clamp(SHA-512(buffer)[0..32)), unreduced; prefix = the other 32 bytes
    pub fn new_key_and_seed_with_input(self, buffer: &[u8]) -> (Scalar, &[u8], Vec<u8>) {
        let mut hasher = Sha512::new();
        hasher.update(buffer);
        let mut digest = hasher.finalize();
        digest[0] &= 0xf8;
        digest[31] &= 0x7f;
        digest[31] |= 0x40;
        let mut secret = self.scalar();
        secret.v.copy_from_slice(&digest[0..32]);
        (secret, buffer, digest[32..].to_vec())
    }

    /// curve.rs:92-101
    pub fn new_key_and_seed<S: crate::cipher::Stream>(self, stream: &mut S) -> Result<(Scalar, Vec<u8>, Vec<u8>), CurveError> {
        let mut buffer = vec![0u8; 32];
        random::bytes(&mut buffer, stream)?;
        let (sc, buff, digest) = self.new_key_and_seed_with_input(&buffer);
        Ok((sc, buff.to_vec(), digest))
    }
}

impl Generator<Scalar> for CurveHip {
    /// curve.rs:104-114
    fn new_key<S: crate::cipher::Stream>(&self, stream: &mut S) -> Result<Option<Scalar>, KeyError> {
        let (secret, _, _) = self.new_key_and_seed(stream)?;
        Ok(Some(secret))
    }
}

impl Default for CurveHip {
    fn default() -> Self {
        CurveHip::new()
    }
}
