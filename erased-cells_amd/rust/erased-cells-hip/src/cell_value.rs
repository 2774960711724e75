//! `CellValue` (src/value.rs:12-271 of the reference): a scalar with a run-time cell type.  Scalar arithmetic
//! stays on the host, as in the reference — one `f64` operation per call; the per-cell form over whole buffers
//! is what runs on the GPU.  This file also holds the 16-byte `ec_value` image used across the ABI.
use crate::error::{check, Error, Result};
use crate::ffi::{ec_value, ec_value_convert};
use crate::{with_ct, CellEncoding, CellType};
use num_traits::{One, ToPrimitive, Zero};

macro_rules! cv_enum {
    ( $(($id:ident, $p:ident)),*) => {
        /// Value variants for each [`CellType`]
        #[derive(Debug, Copy, Clone)]
        pub enum CellValue { $($id($p)),* }
    }
}
with_ct!(cv_enum);

impl CellValue {
    /// Construct new [`CellValue`] from a statically known [`CellEncoding`].
    pub fn new<T: CellEncoding + Sized>(value: T) -> Self {
        value.into_cell_value()
    }

    /// Get the [`CellType`] encoding `self`.
    pub fn cell_type(&self) -> CellType {
        macro_rules! cv_ct {
            ($( ($id:ident, $_p:ident) ),*) => {
                match self {
                    $(CellValue::$id(_) => CellType::$id),*
                }
            };
        }
        with_ct!(cv_ct)
    }

    /// Get the [`CellValue`] contents as a `T`: `Ok(T)` if `T`'s cell type is the same or wider than the
    /// encoded value's, `Err(NarrowingError)` if it is narrower.
    pub fn get<T: CellEncoding>(&self) -> Result<T> {
        let err = || Error::NarrowingError { src: self.cell_type(), dst: T::cell_type() };
        let cv = self.convert(T::cell_type())?;
        macro_rules! conv {
             ($( ($id:ident, $_p:ident) ),*) => {
                 match cv {
                     $(CellValue::$id(v) => T::static_cast(v).ok_or_else(err),)*
                 }
            };
        }
        with_ct!(conv)
    }

    /// Convert `self` into a variant with [`CellType`] `cell_type` equal to or wider than its current one
    /// (the library's host-side `ec_value_convert`: lattice check first, then the `as` cast).
    pub fn convert(&self, cell_type: CellType) -> Result<Self> {
        let (src, mut dst) = (self.to_ffi(), CellValue::UInt8(0).to_ffi());
        check(unsafe { ec_value_convert(&src, cell_type as u8, &mut dst) })?;
        Ok(CellValue::from_ffi(&dst))
    }

    /// Converts both values to the smallest cell-type that can contain `self` and `other`.
    pub fn unify(&self, other: &Self) -> (Self, Self) {
        let dest = self.cell_type().union(other.cell_type());
        // `unwrap` is fine: `a.union(b)` holds both `a` and `b` for every pair of the lattice
        (self.convert(dest).unwrap(), other.convert(dest).unwrap())
    }

    /// The value 0 or 1 (`k`) of cell type `ct` (`CellType::zero` / `CellType::one`).
    pub(crate) fn small(ct: CellType, k: u8) -> Self {
        macro_rules! small {
            ($( ($id:ident, $p:ident) ),*) => {
                match ct {
                    $(CellType::$id => CellValue::$id(k as $p),)*
                }
            };
        }
        with_ct!(small)
    }

    /// `self as f64` (what `to_f64().unwrap()` of src/value.rs:207 yields): exact up to 2^53, round-to-nearest-even above.
    pub(crate) fn as_f64(&self) -> f64 {
        macro_rules! as_f64 {
            ($( ($id:ident, $_p:ident) ),*) => {
                match *self {
                    $(CellValue::$id(v) => v as f64,)*
                }
            };
        }
        with_ct!(as_f64)
    }

    /// The payload as the low bytes of a `u64` (the C union of `ec_value` is 8 little-endian bytes).
    pub(crate) fn bits(&self) -> u64 {
        match *self {
            CellValue::UInt8(v) => v as u64,
            CellValue::UInt16(v) => v as u64,
            CellValue::UInt32(v) => v as u64,
            CellValue::UInt64(v) => v,
            CellValue::Int8(v) => v as u8 as u64,
            CellValue::Int16(v) => v as u16 as u64,
            CellValue::Int32(v) => v as u32 as u64,
            CellValue::Int64(v) => v as u64,
            CellValue::Float32(v) => v.to_bits() as u64,
            CellValue::Float64(v) => v.to_bits(),
        }
    }

    pub(crate) fn from_bits(ct: CellType, b: u64) -> Self {
        match ct {
            CellType::UInt8 => CellValue::UInt8(b as u8),
            CellType::UInt16 => CellValue::UInt16(b as u16),
            CellType::UInt32 => CellValue::UInt32(b as u32),
            CellType::UInt64 => CellValue::UInt64(b),
            CellType::Int8 => CellValue::Int8(b as u8 as i8),
            CellType::Int16 => CellValue::Int16(b as u16 as i16),
            CellType::Int32 => CellValue::Int32(b as u32 as i32),
            CellType::Int64 => CellValue::Int64(b as i64),
            CellType::Float32 => CellValue::Float32(f32::from_bits(b as u32)),
            CellType::Float64 => CellValue::Float64(f64::from_bits(b)),
        }
    }

    pub(crate) fn to_ffi(&self) -> ec_value {
        ec_value { dtype: self.cell_type() as u8, pad_: [0; 7], bits: self.bits() }
    }

    pub(crate) fn from_ffi(v: &ec_value) -> Self {
        Self::from_bits(CellType::from_code(v.dtype), v.bits)
    }

    /// Order-preserving key of a value among values of ITS cell type: integers as themselves, floats by
    /// `total_cmp` (−NaN < −inf < … < −0 < +0 < … < +inf < +NaN).
    fn order_key(&self) -> i128 {
        match *self {
            CellValue::UInt8(v) => v as i128,
            CellValue::UInt16(v) => v as i128,
            CellValue::UInt32(v) => v as i128,
            CellValue::UInt64(v) => v as i128,
            CellValue::Int8(v) => v as i128,
            CellValue::Int16(v) => v as i128,
            CellValue::Int32(v) => v as i128,
            CellValue::Int64(v) => v as i128,
            CellValue::Float32(v) => {
                let b = v.to_bits() as i32;
                (b ^ ((((b >> 31) as u32) >> 1) as i32)) as i128
            }
            CellValue::Float64(v) => {
                let b = v.to_bits() as i64;
                (b ^ ((((b >> 63) as u64) >> 1) as i64)) as i128
            }
        }
    }
}

/// Convert from primitive to [`CellValue`].
impl<T: CellEncoding> From<T> for CellValue {
    fn from(value: T) -> Self {
        value.into_cell_value()
    }
}

/// Provide `num_traits` interop.
impl ToPrimitive for CellValue {
    fn to_i64(&self) -> Option<i64> {
        macro_rules! conv {
            ($( ($id:ident, $_p:ident) ),*) => {
                match self {
                    $(CellValue::$id(v) => v.to_i64(),)*
                }
            }
        }
        with_ct!(conv)
    }

    fn to_u64(&self) -> Option<u64> {
        macro_rules! conv {
            ($( ($id:ident, $_p:ident) ),*) => {
                match self {
                    $(CellValue::$id(v) => v.to_u64(),)*
                }
            }
        }
        with_ct!(conv)
    }

    fn to_f64(&self) -> Option<f64> {
        Some(self.as_f64())
    }
}

impl One for CellValue {
    #[inline]
    fn one() -> Self {
        CellValue::UInt8(1)
    }
}

impl Zero for CellValue {
    #[inline]
    fn zero() -> Self {
        CellValue::UInt8(0)
    }

    fn is_zero(&self) -> bool {
        macro_rules! zero {
             ($( ($id:ident, $_p:ident) ),*) => {
                match self {
                    $(CellValue::$id(v) => v.is_zero(),)*
                }
            }
        }
        with_ct!(zero)
    }
}

pub(crate) mod ops {
    use crate::CellValue;
    use std::cmp::Ordering;
    use std::ops::{Add, Div, Mul, Neg, Sub};

    // Every binary op computes `(l as f64) op (r as f64)` and yields `Float64`, for all 100 operand-type pairs
    // (src/value.rs:199-217): the `unify` in front of the casts is value-preserving, so it is skipped here.
    macro_rules! cv_bin_op {
        ($trt:ident, $mth:ident, $op:tt) => {
            impl <R> $trt<R> for &CellValue where R: Into<CellValue> {
                type Output = CellValue;
                fn $mth(self, rhs: R) -> Self::Output {
                    let rhs: CellValue = rhs.into();
                    CellValue::Float64(self.as_f64() $op rhs.as_f64())
                }
            }
            impl <R> $trt<R> for CellValue where R: Into<CellValue> {
                type Output = CellValue;
                fn $mth(self, rhs: R) -> Self::Output {
                    $trt::$mth(&self, rhs)
                }
            }
        }
    }
    cv_bin_op!(Add, add, +);
    cv_bin_op!(Sub, sub, -);
    cv_bin_op!(Mul, mul, *);
    cv_bin_op!(Div, div, /);

    impl Neg for CellValue {
        type Output = CellValue;
        /// src/value.rs:224-240: u8 -> i16, u16 -> i32, u32/u64 -> f64, signed and float types keep theirs.
        /// A signed MIN wraps, as the reference does in release builds (and as the device kernel does).
        fn neg(self) -> Self::Output {
            match self {
                CellValue::UInt8(v) => CellValue::Int16(-(v as i16)),
                CellValue::UInt16(v) => CellValue::Int32(-(v as i32)),
                CellValue::UInt32(v) => CellValue::Float64(-(v as f64)),
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
            Some(self.cmp(other))
        }
    }

    impl Ord for CellValue {
        /// Unify, then integers by value and floats by `total_cmp` (src/value.rs:248-265).
        fn cmp(&self, other: &Self) -> Ordering {
            let (lhs, rhs) = self.unify(other);
            lhs.order_key().cmp(&rhs.order_key())
        }
    }

    impl PartialEq<Self> for CellValue {
        fn eq(&self, other: &Self) -> bool {
            Ord::cmp(self, other) == Ordering::Equal
        }
    }

    impl Eq for CellValue {}
}
