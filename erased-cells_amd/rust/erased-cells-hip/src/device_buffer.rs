//! [`CellBuffer`] with its cells resident in HBM.
//!
//! PROVENANCE.  The type's name, its method signatures and its trait / operator impl headers are the reference's public
//! surface (erased-cells 0.1.1, src/buffer.rs:12-436, MIT License, Copyright (c) 2023 Astraea, Inc.); the bodies are
//! this crate's.  See INTEGRATION.md §2 for the line ranges.
//!
//! The reference's `Vec<T>`-per-variant enum becomes a cell-type tag, a length and one device allocation, and every
//! per-cell iterator chain of the reference becomes one call into liberased_cells_hip.so.  The host keeps what the
//! reference's host code decides once per operation: the dtype tag, zip truncation (the shorter operand wins), and
//! "collecting nothing gives a `UInt8` buffer" (src/buffer.rs:233-234).
use crate::device::{download, stream, upload, DeviceMem};
use crate::error::{check, must, Error, Result};
use crate::ffi::*;
use crate::{with_ct, BufferOps, CellEncoding, CellType, CellValue, Elided};
use num_traits::ToPrimitive;
use std::cmp::Ordering;
use std::fmt::{Debug, Formatter};
use std::ops::{Add, Div, Mul, Neg, Sub};
use std::os::raw::c_void;

/// Cells of one run-time [`CellType`], in HBM.
pub struct CellBuffer {
    pub(crate) ct: CellType,
    pub(crate) len: usize,
    pub(crate) mem: DeviceMem,
}

impl CellBuffer {
    pub fn new<T: CellEncoding>(data: Vec<T>) -> Self {
        Self::from_host(&data)
    }

    /// `len` cells of type `ct`, contents undefined until a kernel or an upload has written them.
    pub(crate) fn uninit(ct: CellType, len: usize) -> Self {
        CellBuffer { ct, len, mem: DeviceMem::new(len * ct.size_of()) }
    }

    /// The buffer an operator returns when there is nothing to compute.
    pub(crate) fn empty_u8() -> Self {
        Self::uninit(CellType::UInt8, 0)
    }

    pub(crate) fn dev_ptr(&self) -> *const c_void {
        self.mem.ptr()
    }

    /// Device address of cell `index`.
    fn at(&self, index: usize) -> *mut c_void {
        (self.mem.ptr() as usize + index * self.ct.size_of()) as *mut c_void
    }

    fn from_host<T: CellEncoding>(data: &[T]) -> Self {
        let fresh = Self::uninit(T::cell_type(), data.len());
        upload(fresh.mem.ptr(), data);
        fresh
    }

    /// Host copy of cells `[start, start + n)`; `P` must be the buffer's own primitive.
    fn fetch<P: CellEncoding>(&self, start: usize, n: usize) -> Vec<P> {
        assert_eq!(self.ct, P::cell_type(), "a {} buffer holds no {} cells", self.ct, P::cell_type());
        assert!(start + n <= self.len);
        download::<P>(self.at(start), n)
    }

    fn check_index(&self, index: usize) {
        assert!(index < self.len, "index out of bounds: the len is {} but the index is {}", self.len, index);
    }

    /// A copy that is `extra` cells longer (the new cells undefined): the device-side half of `Extend`.
    fn grown_by(&self, extra: usize) -> Self {
        let grown = Self::uninit(self.ct, self.len + extra);
        if self.len > 0 {
            must(unsafe { ec_copy(grown.mem.ptr(), self.dev_ptr(), self.len * self.ct.size_of(), stream()) }, "ec_copy");
        }
        grown
    }

    /// `self op rhs`, cell by cell over the shorter length, always into a Float64 buffer: one `ec_binop` launch.
    pub(crate) fn binop(&self, op: ec_op, rhs: &Self) -> Self {
        let n = usize::min(self.len, rhs.len);
        if n == 0 {
            return Self::empty_u8();
        }
        let result = Self::uninit(CellType::Float64, n);
        must(
            unsafe { ec_binop(op, self.ct as u8, self.dev_ptr(), rhs.ct as u8, rhs.dev_ptr(), n, result.mem.ptr() as *mut f64, stream()) },
            "ec_binop",
        );
        result
    }

    /// `self op scalar`: the scalar crosses the ABI as a tagged value and is widened to f64 once, on the host.
    pub(crate) fn binop_scalar(&self, op: ec_op, rhs: CellValue) -> Self {
        if self.len == 0 {
            return Self::empty_u8();
        }
        let result = Self::uninit(CellType::Float64, self.len);
        let scalar = rhs.to_ffi();
        must(
            unsafe { ec_binop_scalar(op, self.ct as u8, self.dev_ptr(), self.len, &scalar, result.mem.ptr() as *mut f64, stream()) },
            "ec_binop_scalar",
        );
        result
    }

    /// `-self`, in the cell type the reference's scalar negation produces (`ec_neg_result_type`).
    fn negated(&self) -> Self {
        if self.len == 0 {
            return Self::empty_u8();
        }
        let result = Self::uninit(CellType::from_code(unsafe { ec_neg_result_type(self.ct as u8) }), self.len);
        must(unsafe { ec_neg(self.ct as u8, self.dev_ptr(), self.len, result.mem.ptr(), stream()) }, "ec_neg");
        result
    }
}

impl BufferOps for CellBuffer {
    fn from_vec<T: CellEncoding>(data: Vec<T>) -> Self {
        Self::from_host(&data)
    }

    fn with_defaults(len: usize, ct: CellType) -> Self {
        Self::fill(len, ct.zero()) // the `Default` of every cell primitive is its zero
    }

    fn fill(len: usize, value: CellValue) -> Self {
        let filled = Self::uninit(value.cell_type(), len);
        let v = value.to_ffi();
        must(unsafe { ec_fill(filled.ct as u8, filled.mem.ptr(), len, &v, stream()) }, "ec_fill");
        filled
    }

    fn fill_via<T, F>(len: usize, f: F) -> Self
    where
        T: CellEncoding,
        F: Fn(usize) -> T,
    {
        let host: Vec<T> = (0..len).map(f).collect();
        Self::from_host(&host)
    }

    fn len(&self) -> usize {
        self.len
    }

    fn is_empty(&self) -> bool {
        self.len == 0
    }

    fn cell_type(&self) -> CellType {
        self.ct
    }

    fn get(&self, index: usize) -> CellValue {
        self.check_index(index);
        let mut payload = 0u64;
        must(
            unsafe { ec_download(&mut payload as *mut u64 as *mut c_void, self.at(index), self.ct.size_of(), stream()) },
            "ec_download",
        );
        CellValue::from_bits(self.ct, payload)
    }

    fn put(&mut self, idx: usize, value: CellValue) -> Result<()> {
        let payload = value.convert(self.ct)?.bits(); // refused before the bounds check, as in the reference
        self.check_index(idx);
        check(unsafe { ec_upload(self.at(idx), &payload as *const u64 as *const c_void, self.ct.size_of(), stream()) })
    }

    fn convert(&self, cell_type: CellType) -> Result<Self> {
        if cell_type == self.ct {
            return Ok(self.clone());
        }
        if !self.ct.can_fit_into(cell_type) {
            return Err(Error::NarrowingError { src: self.ct, dst: cell_type });
        }
        if self.len == 0 {
            return Ok(Self::empty_u8());
        }
        let widened = Self::uninit(cell_type, self.len);
        check(unsafe { ec_convert(self.ct as u8, self.dev_ptr(), cell_type as u8, widened.mem.ptr(), self.len, stream()) })?;
        Ok(widened)
    }

    fn min_max(&self) -> (CellValue, CellValue) {
        let mut lo = CellValue::UInt8(0).to_ffi();
        let mut hi = lo;
        must(
            unsafe { ec_min_max(self.ct as u8, self.dev_ptr(), std::ptr::null(), self.len, &mut lo, &mut hi, stream()) },
            "ec_min_max",
        );
        (CellValue::from_ffi(&lo), CellValue::from_ffi(&hi))
    }

    fn to_vec<T: CellEncoding>(self) -> Result<Vec<T>> {
        let as_t = self.convert(T::cell_type())?;
        Ok(as_t.fetch::<T>(0, as_t.len)) // `fetch` asserts the cell types agree (an empty convert yields UInt8)
    }
}

impl Clone for CellBuffer {
    fn clone(&self) -> Self {
        self.grown_by(0)
    }
}

impl Debug for CellBuffer {
    /// `UInt8CellBuffer(0, 1, 2, 3, 4, ... 95, 96, 97, 98, 99)`.  More than ten cells: only the first and last five are
    /// downloaded (two small copies), never the buffer.
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        write!(f, "{}CellBuffer(", self.ct)?;
        macro_rules! shown {
            ( $(($id:ident, $p:ident)),* ) => {
                match self.ct {
                    $( CellType::$id if self.len > 10 => {
                        let head = self.fetch::<$p>(0, 5);
                        let tail = self.fetch::<$p>(self.len - 5, 5);
                        write!(f, "{:?}, ... {:?}", Elided(&head), Elided(&tail))?
                    }
                    CellType::$id => write!(f, "{:?}", Elided(&self.fetch::<$p>(0, self.len)))?, )*
                }
            };
        }
        with_ct!(shown);
        f.write_str(")")
    }
}

/// `with_ct!` with a third column: the num-traits method that converts a value INTO that primitive.
macro_rules! with_ct_and_to {
    ($callback:ident) => {
        $callback! {
            (UInt8, u8, to_u8), (UInt16, u16, to_u16), (UInt32, u32, to_u32), (UInt64, u64, to_u64),
            (Int8, i8, to_i8), (Int16, i16, to_i16), (Int32, i32, to_i32), (Int64, i64, to_i64),
            (Float32, f32, to_f32), (Float64, f64, to_f64)
        }
    };
}

impl<C: CellEncoding> Extend<C> for CellBuffer {
    /// Items are converted to the buffer's primitive by VALUE (num-traits' range-checked `to_<p>` of [`CellValue`]), not
    /// by cell type: `300u16` does not fit a `UInt8` buffer and panics, `200u16` does.  The whole batch is converted on
    /// the host and appended with one reallocation, one device copy of the old cells and one upload of the new ones.
    fn extend<T: IntoIterator<Item = C>>(&mut self, iter: T) {
        let incoming: Vec<CellValue> = iter.into_iter().map(CellValue::new).collect();
        macro_rules! append {
            ( $(($id:ident, $p:ident, $to:ident)),* ) => {
                match self.ct {
                    $( CellType::$id => {
                        let fitted: Vec<$p> = incoming
                            .iter()
                            .map(|v| v.$to().unwrap_or_else(|| panic!("{v:?} does not fit into a {} cell", self.ct)))
                            .collect();
                        let grown = self.grown_by(fitted.len());
                        upload(grown.at(self.len), &fitted);
                        *self = grown;
                    } )*
                }
            };
        }
        with_ct_and_to!(append);
    }
}

impl<C: CellEncoding> FromIterator<C> for CellBuffer {
    fn from_iter<T: IntoIterator<Item = C>>(iter: T) -> Self {
        let host: Vec<C> = Vec::from_iter(iter);
        Self::from_host(&host)
    }
}

impl FromIterator<CellValue> for CellBuffer {
    /// The FIRST value decides the cell type (no values: `UInt8`); every value must then widen into it — one that does
    /// not panics, as the reference's `get().unwrap()` does.
    fn from_iter<T: IntoIterator<Item = CellValue>>(iterable: T) -> Self {
        let mut values = iterable.into_iter().peekable();
        let Some(decides) = values.peek().map(CellValue::cell_type) else {
            return CellBuffer::with_defaults(0, CellType::UInt8);
        };
        macro_rules! gather {
            ( $(($id:ident, $p:ident)),* ) => {
                match decides {
                    $( CellType::$id => {
                        let host: Vec<$p> = values.map(|v| v.get::<$p>().expect("a value that fits the first value's cell type")).collect();
                        CellBuffer::from_host(&host)
                    } )*
                }
            };
        }
        with_ct!(gather)
    }
}

impl<T: CellEncoding> From<Vec<T>> for CellBuffer {
    fn from(values: Vec<T>) -> Self {
        CellBuffer::from_host(&values)
    }
}

impl<T: CellEncoding> From<&[T]> for CellBuffer {
    fn from(values: &[T]) -> Self {
        CellBuffer::from_host(values)
    }
}

impl<'buf> IntoIterator for &'buf CellBuffer {
    type Item = CellValue;
    type IntoIter = CellBufferIterator<'buf>;

    /// The whole buffer comes over in ONE download; the iterator then decodes cells from that host image.  (Fetching
    /// cell by cell, as an iterator over `get` would, costs a device round trip per cell.)
    fn into_iter(self) -> Self::IntoIter {
        let image = download::<u8>(self.dev_ptr(), self.len * self.ct.size_of());
        CellBufferIterator { of: self, image, next_cell: 0 }
    }
}

/// Walks the cells of a [`CellBuffer`] as [`CellValue`]s over a host image of the buffer.
pub struct CellBufferIterator<'buf> {
    of: &'buf CellBuffer,
    image: Vec<u8>,
    next_cell: usize,
}

impl Iterator for CellBufferIterator<'_> {
    type Item = CellValue;

    fn next(&mut self) -> Option<Self::Item> {
        let width = self.of.ct.size_of();
        let bytes = self.image.get(self.next_cell * width..(self.next_cell + 1) * width)?;
        let mut payload = [0u8; 8];
        payload[..width].copy_from_slice(bytes);
        self.next_cell += 1;
        Some(CellValue::from_bits(self.of.ct, u64::from_le_bytes(payload)))
    }

    fn size_hint(&self) -> (usize, Option<usize>) {
        let left = self.of.len - self.next_cell.min(self.of.len);
        (left, Some(left))
    }
}

impl<C: CellEncoding> TryFrom<CellBuffer> for Vec<C> {
    type Error = Error;

    fn try_from(value: CellBuffer) -> Result<Self> {
        value.to_vec::<C>()
    }
}

// api-surface(src/buffer.rs:318-436): the sixteen binary-operator impl headers, the two `Neg` impls and the ordering /
// equality impl headers of CellBuffer (`&a op &b`, `a op b`, `a op &b`, `a op scalar`; no `scalar op a` in the reference either)
macro_rules! buffer_operator {
    ($trt:ident, $mth:ident, $code:expr) => {
        impl $trt for &CellBuffer {
            type Output = CellBuffer;
            fn $mth(self, rhs: Self) -> Self::Output {
                self.binop($code, rhs)
            }
        }
        impl $trt for CellBuffer {
            type Output = CellBuffer;
            fn $mth(self, rhs: Self) -> Self::Output {
                self.binop($code, &rhs)
            }
        }
        impl $trt<&CellBuffer> for CellBuffer {
            type Output = CellBuffer;
            fn $mth(self, rhs: &CellBuffer) -> Self::Output {
                self.binop($code, rhs)
            }
        }
        impl<R> $trt<R> for CellBuffer
        where
            R: Into<CellValue>,
        {
            type Output = CellBuffer;
            fn $mth(self, rhs: R) -> Self::Output {
                self.binop_scalar($code, rhs.into())
            }
        }
    };
}
buffer_operator!(Add, add, EC_ADD);
buffer_operator!(Sub, sub, EC_SUB);
buffer_operator!(Mul, mul, EC_MUL);
buffer_operator!(Div, div, EC_DIV);

impl Neg for &CellBuffer {
    type Output = CellBuffer;
    fn neg(self) -> Self::Output {
        self.negated()
    }
}

impl Neg for CellBuffer {
    type Output = CellBuffer;
    fn neg(self) -> Self::Output {
        self.negated()
    }
}

impl PartialEq<Self> for CellBuffer {
    fn eq(&self, other: &Self) -> bool {
        self.cmp(other).is_eq()
    }
}

impl Eq for CellBuffer {}

impl PartialOrd for CellBuffer {
    fn partial_cmp(&self, other: &Self) -> Option<Ordering> {
        Some(Ord::cmp(self, other))
    }
}

impl Ord for CellBuffer {
    /// Cell type first; then the first pair of differing cells decides, under the total order (`total_cmp` for floats,
    /// so equality is bit equality); then length.  Decided on the device (`ec_buffer_cmp`: a first-difference
    /// reduction) — no cell is downloaded.
    fn cmp(&self, other: &Self) -> Ordering {
        let mut sign = 0i32;
        must(
            unsafe { ec_buffer_cmp(self.ct as u8, self.dev_ptr(), self.len, other.ct as u8, other.dev_ptr(), other.len, &mut sign, stream()) },
            "ec_buffer_cmp",
        );
        sign.cmp(&0)
    }
}
// end api-surface
