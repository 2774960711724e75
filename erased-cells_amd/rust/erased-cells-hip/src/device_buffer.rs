//! `CellBuffer` with its cells resident in HBM: the reference's `Vec<T>`-per-variant enum (src/buffer.rs:12-55)
//! becomes a cell-type tag, a length and one device allocation, and every per-cell iterator chain of the
//! reference (`self.into_iter().zip(..).map(..).collect()`) becomes one call into liberased_cells_hip.so.
//! The host keeps what the reference's host code decides once per operation: the dtype tag (`with_ct!`), zip
//! truncation (src/buffer.rs:327), "an empty result is a UInt8 buffer" (src/buffer.rs:233-234).
use crate::device::{download, stream, upload, DeviceMem};
use crate::error::{check, must, Error, Result};
use crate::ffi::*;
use crate::{with_ct, BufferOps, CellEncoding, CellType, CellValue};
use num_traits::ToPrimitive;
use std::fmt::{Debug, Formatter};
use std::os::raw::c_void;

/// A buffer of cells of one run-time [`CellType`], device-resident.
pub struct CellBuffer {
    pub(crate) ct: CellType,
    pub(crate) len: usize,
    pub(crate) mem: DeviceMem,
}

/// `(CellType, primitive, num-traits range-checked conversion)` for all ten encodings.
macro_rules! with_ct_conv {
    ($callback:ident) => {
        $callback! {
            (UInt8, u8, to_u8), (UInt16, u16, to_u16), (UInt32, u32, to_u32), (UInt64, u64, to_u64),
            (Int8, i8, to_i8), (Int16, i16, to_i16), (Int32, i32, to_i32), (Int64, i64, to_i64),
            (Float32, f32, to_f32), (Float64, f64, to_f64)
        }
    };
}

impl CellBuffer {
    pub fn new<T: CellEncoding>(data: Vec<T>) -> Self {
        data.into()
    }

    pub(crate) fn uninit(ct: CellType, len: usize) -> Self {
        Self { ct, len, mem: DeviceMem::new(len * ct.size_of()) }
    }
    /// What `collect()` of nothing yields in the reference (src/buffer.rs:233-234).
    pub(crate) fn empty_u8() -> Self {
        Self::uninit(CellType::UInt8, 0)
    }
    pub(crate) fn dev_ptr(&self) -> *const c_void {
        self.mem.ptr()
    }
    fn cell_ptr(&self, index: usize) -> *mut c_void {
        unsafe { (self.mem.ptr() as *mut u8).add(index * self.ct.size_of()) as *mut c_void }
    }
    fn upload_slice<T: CellEncoding>(data: &[T]) -> Self {
        let out = Self::uninit(T::cell_type(), data.len());
        upload(out.mem.ptr(), data);
        out
    }
    /// Cells `[start, start + n)` as host values of the buffer's own primitive type `P`.
    fn download_cells<P: CellEncoding>(&self, start: usize, n: usize) -> Vec<P> {
        assert_eq!(self.ct, P::cell_type());
        assert!(start + n <= self.len);
        download::<P>(self.cell_ptr(start), n)
    }

    pub(crate) fn binop(&self, op: ec_op, rhs: &Self) -> Self {
        let n = self.len.min(rhs.len); // zip (src/buffer.rs:327)
        if n == 0 {
            return Self::empty_u8();
        }
        let out = Self::uninit(CellType::Float64, n); // every binop widens to f64 (src/value.rs:207)
        must(
            unsafe { ec_binop(op, self.ct as u8, self.dev_ptr(), rhs.ct as u8, rhs.dev_ptr(), n, out.mem.ptr() as *mut f64, stream()) },
            "ec_binop",
        );
        out
    }
    pub(crate) fn binop_scalar(&self, op: ec_op, rhs: CellValue) -> Self {
        if self.len == 0 {
            return Self::empty_u8();
        }
        let out = Self::uninit(CellType::Float64, self.len);
        let v = rhs.to_ffi();
        must(
            unsafe { ec_binop_scalar(op, self.ct as u8, self.dev_ptr(), self.len, &v, out.mem.ptr() as *mut f64, stream()) },
            "ec_binop_scalar",
        );
        out
    }
}

impl BufferOps for CellBuffer {
    /// `From<Vec<T>>`: one host-to-HBM copy.
    fn from_vec<T: CellEncoding>(data: Vec<T>) -> Self {
        data.into()
    }

    fn with_defaults(len: usize, ct: CellType) -> Self {
        Self::fill(len, ct.zero()) // `$p::default()` of every primitive is its zero
    }

    fn fill(len: usize, value: CellValue) -> Self {
        let out = Self::uninit(value.cell_type(), len);
        let v = value.to_ffi();
        must(unsafe { ec_fill(out.ct as u8, out.mem.ptr(), len, &v, stream()) }, "ec_fill");
        out
    }

    fn fill_via<T, F>(len: usize, f: F) -> Self
    where
        T: CellEncoding,
        F: Fn(usize) -> T,
    {
        let v: Vec<T> = (0..len).map(f).collect();
        Self::from_vec(v)
    }

    fn len(&self) -> usize {
        self.len
    }

    fn is_empty(&self) -> bool {
        self.len() == 0
    }

    fn cell_type(&self) -> CellType {
        self.ct
    }

    /// Panics when `index` is out of bounds, as `Vec` indexing does in the reference.
    fn get(&self, index: usize) -> CellValue {
        assert!(index < self.len, "index out of bounds: the len is {} but the index is {}", self.len, index);
        let mut bits = 0u64;
        must(
            unsafe { ec_download(&mut bits as *mut u64 as *mut c_void, self.cell_ptr(index), self.ct.size_of(), stream()) },
            "ec_download",
        );
        CellValue::from_bits(self.ct, bits)
    }

    fn put(&mut self, idx: usize, value: CellValue) -> Result<()> {
        let value = value.convert(self.cell_type())?;
        assert!(idx < self.len, "index out of bounds: the len is {} but the index is {}", self.len, idx);
        let bits = value.bits();
        check(unsafe { ec_upload(self.cell_ptr(idx), &bits as *const u64 as *const c_void, self.ct.size_of(), stream()) })
    }

    fn convert(&self, cell_type: CellType) -> Result<Self> {
        if cell_type == self.cell_type() {
            return Ok(self.clone());
        }
        if !self.cell_type().can_fit_into(cell_type) {
            return Err(Error::NarrowingError { src: self.cell_type(), dst: cell_type });
        }
        if self.len == 0 {
            return Ok(Self::empty_u8()); // `collect()` of no cells
        }
        let out = Self::uninit(cell_type, self.len);
        check(unsafe { ec_convert(self.ct as u8, self.dev_ptr(), cell_type as u8, out.mem.ptr(), self.len, stream()) })?;
        Ok(out)
    }

    /// Total order (integers natural, floats `total_cmp`), folded from `(T::MAX, T::MIN)` — finite for floats.
    fn min_max(&self) -> (CellValue, CellValue) {
        let (mut mn, mut mx) = (CellValue::UInt8(0).to_ffi(), CellValue::UInt8(0).to_ffi());
        must(
            unsafe { ec_min_max(self.ct as u8, self.dev_ptr(), std::ptr::null(), self.len, &mut mn, &mut mx, stream()) },
            "ec_min_max",
        );
        (CellValue::from_ffi(&mn), CellValue::from_ffi(&mx))
    }

    fn to_vec<T: CellEncoding>(self) -> Result<Vec<T>> {
        let r = self.convert(T::cell_type())?;
        Ok(r.download_cells::<T>(0, r.len)) // asserts the cell types agree, as `danger::cast` does
    }
}

impl Clone for CellBuffer {
    fn clone(&self) -> Self {
        let out = Self::uninit(self.ct, self.len);
        if self.len > 0 {
            must(unsafe { ec_copy(out.mem.ptr(), self.dev_ptr(), self.len * self.ct.size_of(), stream()) }, "ec_copy");
        }
        out
    }
}

impl Debug for CellBuffer {
    /// `UInt8CellBuffer(0, 1, 2, 3, 4, ... 95, 96, 97, 98, 99)`: only the cells that are shown are downloaded.
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        use crate::Elided;
        let basename = self.cell_type().to_string();
        macro_rules! render {
            ( $(($id:ident, $p:ident)),*) => {{
                f.write_fmt(format_args!("{basename}CellBuffer("))?;
                match self.ct {
                    $(CellType::$id => {
                        if self.len > 10 {
                            let (head, tail) = (self.download_cells::<$p>(0, 5), self.download_cells::<$p>(self.len - 5, 5));
                            f.write_fmt(format_args!("{:?}, ... {:?}", Elided(&head), Elided(&tail)))?
                        } else {
                            f.write_fmt(format_args!("{:?}", Elided(&self.download_cells::<$p>(0, self.len))))?
                        }
                    })*
                };
                f.write_str(")")
            }}
        }
        with_ct!(render)
    }
}

impl<C: CellEncoding> Extend<C> for CellBuffer {
    /// Each item goes through num-traits' range-checked `to_<p>()` (value-based, unlike `convert`) and panics
    /// when it does not fit the buffer's cell type (src/buffer.rs:205-221); the buffer grows by one
    /// reallocation + device copy per call.
    fn extend<T: IntoIterator<Item = C>>(&mut self, iter: T) {
        macro_rules! grow {
            ( $(($id:ident, $p:ident, $conv:ident)),*) => {
                match self.ct {
                    $(CellType::$id => {
                        let tail: Vec<$p> = iter.into_iter().map(|c| c.into_cell_value().$conv().unwrap()).collect();
                        let grown = CellBuffer::uninit(self.ct, self.len + tail.len());
                        if self.len > 0 {
                            must(unsafe { ec_copy(grown.mem.ptr(), self.dev_ptr(), self.len * self.ct.size_of(), stream()) }, "ec_copy");
                        }
                        upload(grown.cell_ptr(self.len), &tail);
                        *self = grown;
                    },)*
                }
            }
        }
        with_ct_conv!(grow);
    }
}

impl<C: CellEncoding> FromIterator<C> for CellBuffer {
    fn from_iter<T: IntoIterator<Item = C>>(iter: T) -> Self {
        Self::from_vec(iter.into_iter().collect())
    }
}

impl FromIterator<CellValue> for CellBuffer {
    /// Empty -> `UInt8`; otherwise the FIRST value's cell type, every value through `get::<T>().unwrap()`.
    fn from_iter<T: IntoIterator<Item = CellValue>>(iterable: T) -> Self {
        let values = iterable.into_iter().collect::<Vec<CellValue>>();
        match values.as_slice() {
            [] => CellBuffer::with_defaults(0, CellType::UInt8),
            [x, ..] => {
                macro_rules! conv {
                    ( $(($id:ident, $p:ident)),*) => {
                        match x.cell_type() {
                            $(CellType::$id => {
                                CellBuffer::from_vec(values.iter().map(|v| v.get::<$p>().unwrap()).collect::<Vec<$p>>())
                            })*
                        }
                    }
                }
                with_ct!(conv)
            }
        }
    }
}

impl<T: CellEncoding> From<Vec<T>> for CellBuffer {
    fn from(values: Vec<T>) -> Self {
        CellBuffer::upload_slice(&values)
    }
}

impl<T: CellEncoding> From<&[T]> for CellBuffer {
    fn from(values: &[T]) -> Self {
        CellBuffer::upload_slice(values)
    }
}

impl<'buf> IntoIterator for &'buf CellBuffer {
    type Item = CellValue;
    type IntoIter = CellBufferIterator<'buf>;
    /// One download for the whole walk (the reference's iterator calls `get` per cell; here that would be a
    /// device round trip per cell).
    fn into_iter(self) -> Self::IntoIter {
        let host = download::<u8>(self.dev_ptr(), self.len * self.ct.size_of());
        CellBufferIterator { buf: self, idx: 0, len: self.len, host }
    }
}

/// Iterator over [`CellValue`] elements in a [`CellBuffer`].
pub struct CellBufferIterator<'buf> {
    buf: &'buf CellBuffer,
    idx: usize,
    len: usize,
    host: Vec<u8>,
}

impl Iterator for CellBufferIterator<'_> {
    type Item = CellValue;

    fn next(&mut self) -> Option<Self::Item> {
        if self.idx >= self.len {
            None
        } else {
            let sz = self.buf.ct.size_of();
            let mut raw = [0u8; 8];
            raw[..sz].copy_from_slice(&self.host[self.idx * sz..(self.idx + 1) * sz]);
            self.idx += 1;
            Some(CellValue::from_bits(self.buf.ct, u64::from_le_bytes(raw)))
        }
    }
}

impl<C: CellEncoding> TryFrom<CellBuffer> for Vec<C> {
    type Error = Error;

    fn try_from(value: CellBuffer) -> Result<Self> {
        value.to_vec()
    }
}

mod ops {
    use crate::error::must;
    use crate::ffi::*;
    use crate::{device::stream, BufferOps, CellBuffer, CellType, CellValue};
    use std::cmp::Ordering;
    use std::ops::{Add, Div, Mul, Neg, Sub};

    // cb_bin_op! (src/buffer.rs:321-358): the iterator-chain bodies become one FFI call.
    macro_rules! cb_bin_op {
        ($trt:ident, $mth:ident, $op:expr) => {
            // Both borrows.
            impl $trt for &CellBuffer {
                type Output = CellBuffer;
                fn $mth(self, rhs: Self) -> Self::Output {
                    self.binop($op, rhs)
                }
            }
            // Both owned/consumed
            impl $trt for CellBuffer {
                type Output = CellBuffer;
                fn $mth(self, rhs: Self) -> Self::Output {
                    $trt::$mth(&self, &rhs)
                }
            }
            // RHS borrow
            impl $trt<&CellBuffer> for CellBuffer {
                type Output = CellBuffer;
                fn $mth(self, rhs: &CellBuffer) -> Self::Output {
                    $trt::$mth(&self, rhs)
                }
            }
            // RHS scalar
            impl <R> $trt<R> for CellBuffer where R: Into<CellValue> {
                type Output = CellBuffer;
                fn $mth(self, rhs: R) -> Self::Output {
                    let r: CellValue = rhs.into();
                    self.binop_scalar($op, r)
                }
            }
        }
    }
    cb_bin_op!(Add, add, EC_ADD);
    cb_bin_op!(Sub, sub, EC_SUB);
    cb_bin_op!(Mul, mul, EC_MUL);
    cb_bin_op!(Div, div, EC_DIV);

    impl Neg for &CellBuffer {
        type Output = CellBuffer;
        /// The result variant widens per src/value.rs:224-240 (u8 -> i16, u16 -> i32, u32/u64 -> f64).
        fn neg(self) -> Self::Output {
            if self.len == 0 {
                return CellBuffer::empty_u8();
            }
            let out = CellBuffer::uninit(CellType::from_code(unsafe { ec_neg_result_type(self.ct as u8) }), self.len);
            must(unsafe { ec_neg(self.ct as u8, self.dev_ptr(), self.len, out.mem.ptr(), stream()) }, "ec_neg");
            out
        }
    }
    impl Neg for CellBuffer {
        type Output = CellBuffer;
        fn neg(self) -> Self::Output {
            Neg::neg(&self)
        }
    }

    impl PartialEq<Self> for CellBuffer {
        fn eq(&self, other: &Self) -> bool {
            Ord::cmp(self, other) == Ordering::Equal
        }
    }

    impl Eq for CellBuffer {}

    impl PartialOrd for CellBuffer {
        fn partial_cmp(&self, other: &Self) -> Option<Ordering> {
            Some(self.cmp(other))
        }
    }

    /// Cell type first, then the first differing cell under the total order (`total_cmp` for floats), then length
    /// (src/buffer.rs:373-436) — found on the device, nothing is downloaded.
    impl Ord for CellBuffer {
        fn cmp(&self, other: &Self) -> Ordering {
            let mut o = 0i32;
            must(
                unsafe {
                    ec_buffer_cmp(self.ct as u8, self.dev_ptr(), self.len(), other.ct as u8, other.dev_ptr(), other.len(), &mut o, stream())
                },
                "ec_buffer_cmp",
            );
            o.cmp(&0)
        }
    }
}
