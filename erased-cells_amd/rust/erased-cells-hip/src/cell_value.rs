//! [`CellValue`]: one scalar with a run-time cell type.
//!
//! PROVENANCE.  The enum shape (`with_ct!(cv_enum)`), the public method signatures and the operator / trait impl
//! headers are the reference's public surface (erased-cells 0.1.1, src/value.rs:12-271, MIT License, Copyright (c) 2023
//! Astraea, Inc.); the bodies are this crate's.  See INTEGRATION.md §2 for the line ranges.
//!
//! Scalar arithmetic stays on the host, as in the reference — one `f64` operation per call; the per-cell form over whole
//! buffers is what runs on the GPU.  A `CellValue` crosses the ABI as the 16-byte `ec_value` (tag + 8 payload bytes),
//! so most of this file works on that image: the payload as a `u64` (`bits`), and the type lattice asked of the library
//! (`ec_value_convert`) rather than restated here.
use crate::error::{check, Error, Result};
use crate::ffi::{ec_value, ec_value_convert};
use crate::{with_ct, CellEncoding, CellType};
use num_traits::{One, ToPrimitive, Zero};
use std::cmp::Ordering;
use std::ops::{Add, Div, Mul, Neg, Sub};

// api-surface(src/value.rs:11-20): the enum, one variant per cell type in `with_ct!` order
macro_rules! cv_enum {
    ( $(($id:ident, $p:ident)),*) => {
        /// A cell outside any buffer: the primitive together with its [`CellType`].
        #[derive(Debug, Copy, Clone)]
        pub enum CellValue { $($id($p)),* }
    }
}
with_ct!(cv_enum);
// end api-surface

/// `on_payload!(value, v => expr)`: `expr` with `v` bound to the primitive inside `value`, whichever variant it is
/// (the ten arms written out once, here, instead of one `with_ct!` callback per method).
macro_rules! on_payload {
    ($value:expr, $v:ident => $body:expr) => {
        match $value {
            CellValue::UInt8($v) => $body,
            CellValue::UInt16($v) => $body,
            CellValue::UInt32($v) => $body,
            CellValue::UInt64($v) => $body,
            CellValue::Int8($v) => $body,
            CellValue::Int16($v) => $body,
            CellValue::Int32($v) => $body,
            CellValue::Int64($v) => $body,
            CellValue::Float32($v) => $body,
            CellValue::Float64($v) => $body,
        }
    };
}

impl CellValue {
    /// Wrap a primitive.  (The cell type is `T`'s own: this is `T::into_cell_value`.)
    pub fn new<T: CellEncoding + Sized>(value: T) -> Self {
        value.into_cell_value()
    }

    /// The tag.
    pub fn cell_type(&self) -> CellType {
        CellType::from_code(self.to_ffi().dtype)
    }

    /// The payload as a `T`: widened if `T`'s cell type is wider, `Err(NarrowingError)` if it is narrower.
    pub fn get<T: CellEncoding>(&self) -> Result<T> {
        let widened = self.convert(T::cell_type())?;
        // after `convert` the payload IS a `T`; `static_cast` only succeeds for the variant that holds one
        on_payload!(widened, v => T::static_cast(v)).ok_or(Error::NarrowingError { src: self.cell_type(), dst: T::cell_type() })
    }

    /// The same number as a value of `cell_type`.  Legal exactly when `self.cell_type().can_fit_into(cell_type)`; the
    /// library's host-side `ec_value_convert` checks the lattice and performs the cast (Rust's `as` for every legal
    /// pair: exact except `u64`/`i64` to `f64`, which round to nearest even).
    pub fn convert(&self, cell_type: CellType) -> Result<Self> {
        let from = self.to_ffi();
        let mut to = from;
        check(unsafe { ec_value_convert(&from, cell_type as u8, &mut to) }).map(|()| CellValue::from_ffi(&to))
    }

    /// Both values in the narrowest cell type that holds both.
    pub fn unify(&self, other: &Self) -> (Self, Self) {
        let common = CellType::union(self.cell_type(), other.cell_type());
        let lift = |v: &Self| v.convert(common).expect("a union holds both of its operands");
        (lift(self), lift(other))
    }

    /// 0 or 1 as a value of cell type `ct` (behind `CellType::zero` / `CellType::one`).
    pub(crate) fn small(ct: CellType, k: u8) -> Self {
        // u8 widens into every unsigned and float type, i8 into every signed integer type
        let seed = if ct.is_integral() && ct.is_signed() { CellValue::Int8(k as i8) } else { CellValue::UInt8(k) };
        seed.convert(ct).expect("0 and 1 exist in every cell type")
    }

    /// The number as an `f64`: exact up to 2^53, round-to-nearest-even above (`as`).
    pub(crate) fn as_f64(&self) -> f64 {
        on_payload!(*self, v => v as f64)
    }

    /// The payload bytes, little-endian, in the low end of a `u64` (the C union of `ec_value`).
    pub(crate) fn bits(&self) -> u64 {
        let mut raw = [0u8; 8];
        on_payload!(*self, v => {
            let bytes = v.to_le_bytes();
            raw[..bytes.len()].copy_from_slice(&bytes);
        });
        u64::from_le_bytes(raw)
    }

    /// Inverse of [`CellValue::bits`] for a known cell type.
    pub(crate) fn from_bits(ct: CellType, b: u64) -> Self {
        let raw = b.to_le_bytes();
        macro_rules! rebuild {
            ( $( ($id:ident, $p:ident) ),* ) => {
                match ct {
                    $( CellType::$id => {
                        let mut own = [0u8; std::mem::size_of::<$p>()];
                        own.copy_from_slice(&raw[..std::mem::size_of::<$p>()]);
                        CellValue::$id(<$p>::from_le_bytes(own))
                    } )*
                }
            };
        }
        with_ct!(rebuild)
    }

    pub(crate) fn to_ffi(&self) -> ec_value {
        let tag = on_payload!(*self, v => cell_type_of(&v)) as u8;
        ec_value { dtype: tag, pad_: [0; 7], bits: self.bits() }
    }

    pub(crate) fn from_ffi(v: &ec_value) -> Self {
        Self::from_bits(CellType::from_code(v.dtype), v.bits)
    }

    /// A key whose integer order is the reference's total order among values of ONE cell type: integers as they are,
    /// floats as `total_cmp` orders them (sign-magnitude bits folded into two's complement: -NaN < -inf < ... < -0 < +0
    /// < ... < +inf < +NaN).
    fn rank(&self) -> i128 {
        match *self {
            CellValue::Float32(v) => {
                let b = v.to_bits() as i32;
                i128::from(b ^ (((b >> 31) as u32 >> 1) as i32))
            }
            CellValue::Float64(v) => {
                let b = v.to_bits() as i64;
                i128::from(b ^ (((b >> 63) as u64 >> 1) as i64))
            }
            CellValue::UInt8(v) => i128::from(v),
            CellValue::UInt16(v) => i128::from(v),
            CellValue::UInt32(v) => i128::from(v),
            CellValue::UInt64(v) => i128::from(v),
            CellValue::Int8(v) => i128::from(v),
            CellValue::Int16(v) => i128::from(v),
            CellValue::Int32(v) => i128::from(v),
            CellValue::Int64(v) => i128::from(v),
        }
    }
}

fn cell_type_of<T: CellEncoding>(_: &T) -> CellType {
    T::cell_type()
}

impl<T: CellEncoding> From<T> for CellValue {
    fn from(value: T) -> Self {
        CellValue::new(value)
    }
}

/// `num_traits` interop: the three required methods delegate to the primitive's own range-checked conversions (the
/// narrower `to_<p>` of the trait derive from these), except that `to_f64` cannot fail for any cell.
impl ToPrimitive for CellValue {
    fn to_i64(&self) -> Option<i64> {
        on_payload!(*self, v => ToPrimitive::to_i64(&v))
    }

    fn to_u64(&self) -> Option<u64> {
        on_payload!(*self, v => ToPrimitive::to_u64(&v))
    }

    fn to_f64(&self) -> Option<f64> {
        Some(self.as_f64())
    }
}

impl One for CellValue {
    #[inline]
    fn one() -> Self {
        CellType::UInt8.one()
    }
}

impl Zero for CellValue {
    #[inline]
    fn zero() -> Self {
        CellType::UInt8.zero()
    }

    /// Zero of any cell type, `-0.0` included (all payload bits clear once the sign of a float is set aside).
    fn is_zero(&self) -> bool {
        self.as_f64() == 0.0
    }
}

// api-surface(src/value.rs:196-271): operator and ordering impl headers of CellValue
// Arithmetic: for all 100 pairs of operand cell types the reference computes `(l as f64) op (r as f64)` and returns a
// Float64 (its `unify` in front of the casts changes no value: SURVEY App. A.1), so the operands go to f64 directly.
macro_rules! value_operator {
    ($trt:ident, $mth:ident, $f:expr) => {
        impl<R> $trt<R> for &CellValue
        where
            R: Into<CellValue>,
        {
            type Output = CellValue;
            fn $mth(self, rhs: R) -> Self::Output {
                let f: fn(f64, f64) -> f64 = $f;
                CellValue::Float64(f(self.as_f64(), rhs.into().as_f64()))
            }
        }
        impl<R> $trt<R> for CellValue
        where
            R: Into<CellValue>,
        {
            type Output = CellValue;
            fn $mth(self, rhs: R) -> Self::Output {
                <&CellValue as $trt<R>>::$mth(&self, rhs)
            }
        }
    };
}
value_operator!(Add, add, |a, b| a + b);
value_operator!(Sub, sub, |a, b| a - b);
value_operator!(Mul, mul, |a, b| a * b);
value_operator!(Div, div, |a, b| a / b);

impl Neg for CellValue {
    type Output = CellValue;
    /// Unsigned cells need room for the sign: u8 -> i16, u16 -> i32, u32 and u64 -> f64.  Signed and float cells keep
    /// their type; a signed MIN wraps, as the reference does in release builds and as the device kernel does.
    fn neg(self) -> Self::Output {
        match self {
            CellValue::UInt8(v) => CellValue::Int16(-i16::from(v)),
            CellValue::UInt16(v) => CellValue::Int32(-i32::from(v)),
            CellValue::UInt32(v) => CellValue::Float64(-f64::from(v)),
            CellValue::UInt64(v) => CellValue::Float64(-(v as f64)),
            CellValue::Int8(v) => CellValue::Int8(v.wrapping_neg()),
            CellValue::Int16(v) => CellValue::Int16(v.wrapping_neg()),
            CellValue::Int32(v) => CellValue::Int32(v.wrapping_neg()),
            CellValue::Int64(v) => CellValue::Int64(v.wrapping_neg()),
            CellValue::Float32(v) => CellValue::Float32(-v),
            CellValue::Float64(v) => CellValue::Float64(-v),
        }
    }
}

impl PartialOrd for CellValue {
    fn partial_cmp(&self, other: &Self) -> Option<Ordering> {
        Some(Ord::cmp(self, other))
    }
}

impl Ord for CellValue {
    /// Values of different cell types are unified first; then integers compare by value and floats by `total_cmp`.
    fn cmp(&self, other: &Self) -> Ordering {
        let (a, b) = self.unify(other);
        a.rank().cmp(&b.rank())
    }
}

impl PartialEq<Self> for CellValue {
    fn eq(&self, other: &Self) -> bool {
        self.cmp(other).is_eq()
    }
}

impl Eq for CellValue {}
// end api-surface
