//! `SuiteEd25519Hip` — the cipher suite protocol code is instantiated with (mirror of
//! src/group/edwards25519/suite.rs:24-140): same hash (SHA-256), XOF (BLAKE3) and randomness, `CurveHip` as the group.
use core::fmt::{Display, Formatter};
use core::ops::{Deref, DerefMut};

use serde::{Deserialize, Serialize};
use sha2::Sha256;

use crate::cipher::Stream;
use crate::group::edwards25519::Scalar;
use crate::group::{Group, HashFactory};
use crate::share::vss::suite::Suite;
use crate::sign::dss;
use crate::util;
use crate::util::key::{Generator, KeyError, Suite as KeySuite};
use crate::{xof, Random, XOFFactory};

use super::{CurveHip, Point};

#[derive(Clone, Copy, Debug, Default, Serialize, Deserialize)]
pub struct SuiteEd25519Hip {
    curve: CurveHip,
}

impl SuiteEd25519Hip {
    /// suite.rs:33-35
    pub fn new_blake3_sha256_ed25519() -> SuiteEd25519Hip {
        SuiteEd25519Hip::default()
    }
}

impl Deref for SuiteEd25519Hip {
    type Target = CurveHip;
    fn deref(&self) -> &Self::Target {
        &self.curve
    }
}
impl DerefMut for SuiteEd25519Hip {
    fn deref_mut(&mut self) -> &mut Self::Target {
        &mut self.curve
    }
}

impl Generator<Scalar> for SuiteEd25519Hip {
    fn new_key<S: crate::cipher::Stream>(&self, stream: &mut S) -> Result<Option<Scalar>, KeyError> {
        self.curve.new_key(stream)
    }
}

impl Display for SuiteEd25519Hip {
    fn fmt(&self, f: &mut Formatter<'_>) -> core::fmt::Result {
        write!(f, "{}", self.curve)
    }
}

impl Group for SuiteEd25519Hip {
    type POINT = Point;

    fn scalar(&self) -> Scalar {
        self.curve.scalar()
    }
    fn scalar_len(&self) -> usize {
        self.curve.scalar_len()
    }
    fn point(&self) -> Point {
        self.curve.point()
    }
    fn point_len(&self) -> usize {
        self.curve.point_len()
    }
    fn is_prime_order(&self) -> Option<bool> {
        self.curve.is_prime_order()
    }
}

// `Dh` comes from the blanket `impl<T: HashFactory> Dh for T` (dh/dh_impl.rs:181), exactly as for SuiteEd25519.

impl Random for SuiteEd25519Hip {
    fn random_stream(&self) -> Box<dyn Stream> {
        Box::<util::random::random_stream::RandStream>::default() // suite.rs:117-125
    }
}

impl XOFFactory for SuiteEd25519Hip {
    fn xof(&self, key: Option<&[u8]>) -> Box<dyn crate::XOF> {
        Box::new(xof::blake3::Xof::new(key)) // suite.rs:127-131
    }
}

impl HashFactory for SuiteEd25519Hip {
    type T = Sha256; // suite.rs:133-135
}

impl Suite for SuiteEd25519Hip {} // share::vss (and through it share::dkg): suite.rs:137
impl dss::Suite for SuiteEd25519Hip {} // sign::dss: suite.rs:138
impl KeySuite for SuiteEd25519Hip {} // util::key::Pair
