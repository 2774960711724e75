//! `CellType` (src/ctype.rs:11-180 of the reference) and `CellEncoding` (src/encoding.rs:9-40): the ten cell encodings
//! and the primitives that carry them.  `self as u8` IS the ABI dtype
//! code; the lattice (`union`, `can_fit_into`, limits) is answered by the library's host-side table, the same
//! one its kernel dispatch uses.
use crate::error::Error;
use crate::ffi::*;
use crate::CellValue;
use num_traits::{One, Zero};
use std::fmt::{Debug, Display, Formatter};
use std::str::FromStr;

/// Cell-type variants, in the order of `with_ct!`.
#[derive(Debug, Copy, Clone, PartialEq, Eq, PartialOrd, Ord, Hash)]
#[repr(u8)]
pub enum CellType {
    UInt8 = 0,
    UInt16 = 1,
    UInt32 = 2,
    UInt64 = 3,
    Int8 = 4,
    Int16 = 5,
    Int32 = 6,
    Int64 = 7,
    Float32 = 8,
    Float64 = 9,
}

const ALL: [CellType; 10] = [
    CellType::UInt8, CellType::UInt16, CellType::UInt32, CellType::UInt64, CellType::Int8, CellType::Int16,
    CellType::Int32, CellType::Int64, CellType::Float32, CellType::Float64,
];

/// `Display` is the same as `Debug`.
impl Display for CellType {
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        Debug::fmt(self, f)
    }
}

impl FromStr for CellType {
    type Err = Error;

    fn from_str(s: &str) -> Result<Self, Self::Err> {
        ALL.iter().copied().find(|ct| ct.to_string() == s).ok_or_else(|| Error::ParseError(s.into(), "CellType"))
    }
}

impl CellType {
    pub(crate) fn from_code(c: u8) -> Self {
        assert!((c as usize) < ALL.len(), "dtype code {c} out of range");
        ALL[c as usize]
    }

    /// Get an iterator over all the valid enumeration values.
    pub fn iter() -> impl Iterator<Item = CellType> {
        ALL.into_iter()
    }

    /// Determine if `self` is integral or floating-point.
    pub fn is_integral(&self) -> bool {
        !matches!(self, CellType::Float32 | CellType::Float64)
    }

    /// Determine if `self` is signed or unsigned.
    pub fn is_signed(&self) -> bool {
        !matches!(self, CellType::UInt8 | CellType::UInt16 | CellType::UInt32 | CellType::UInt64)
    }

    /// Number of bytes needed to encode `self`.
    pub fn size_of(&self) -> usize {
        unsafe { ec_size_of(*self as u8) }
    }

    /// Select the `CellType` that can numerically contain both `self` and `other`.
    pub fn union(self, other: Self) -> Self {
        Self::from_code(unsafe { ec_union(self as u8, other as u8) })
    }

    /// Determine of `self` can fit within `other`.
    pub fn can_fit_into(self, other: Self) -> bool {
        unsafe { ec_can_fit_into(self as u8, other as u8) != 0 }
    }

    /// Construct the zero value for a variant.
    pub fn zero(&self) -> CellValue {
        CellValue::small(*self, 0)
    }

    /// Construct the one value for a variant.
    pub fn one(&self) -> CellValue {
        CellValue::small(*self, 1)
    }

    /// Determine the minimum value that can be represented by `self`.
    pub fn min_value(&self) -> CellValue {
        let mut v = CellValue::UInt8(0).to_ffi();
        unsafe { ec_min_value(*self as u8, &mut v) };
        CellValue::from_ffi(&v)
    }

    /// Determine the maximum value that can be represented by `self`.
    pub fn max_value(&self) -> CellValue {
        let mut v = CellValue::UInt8(0).to_ffi();
        unsafe { ec_max_value(*self as u8, &mut v) };
        CellValue::from_ffi(&v)
    }
}

/// Trait for marking Rust primitives as having a corresponding [`CellType`]: implemented for exactly the ten
/// primitives a cell can hold (`u8` … `f64`); `isize`, `bool`, `u128` … are not cell encodings.
pub trait CellEncoding: Copy + Debug + Default + Zero + One + PartialEq {
    /// Returns the [`CellType`] covering `Self`.
    fn cell_type() -> CellType;
    /// Converts `self` into a [`CellValue`].
    fn into_cell_value(self) -> CellValue;
    /// Convert dynamic type to static type when logically known: `Some` only when `T` is exactly `Self` (equal cell
    /// types mean the same primitive, so the value is copied bit for bit), `None` for every other pair — there is
    /// no numeric conversion here, that is `CellValue::convert`'s job.
    fn static_cast<T: CellEncoding + Sized>(value: T) -> Option<Self> {
        (Self::cell_type() == T::cell_type()).then(|| {
            debug_assert_eq!(std::mem::size_of::<T>(), std::mem::size_of::<Self>());
            let mut same = Self::default();
            unsafe {
                std::ptr::copy_nonoverlapping(&value as *const T as *const u8, &mut same as *mut Self as *mut u8, std::mem::size_of::<Self>())
            };
            same
        })
    }
}

macro_rules! cell_encoding_of {
    ($prim:ty => $ct:ident) => {
        impl CellEncoding for $prim {
            fn cell_type() -> CellType {
                CellType::$ct
            }
            fn into_cell_value(self) -> CellValue {
                CellValue::$ct(self)
            }
        }
    };
}
cell_encoding_of!(u8 => UInt8);
cell_encoding_of!(u16 => UInt16);
cell_encoding_of!(u32 => UInt32);
cell_encoding_of!(u64 => UInt64);
cell_encoding_of!(i8 => Int8);
cell_encoding_of!(i16 => Int16);
cell_encoding_of!(i32 => Int32);
cell_encoding_of!(i64 => Int64);
cell_encoding_of!(f32 => Float32);
cell_encoding_of!(f64 => Float64);
