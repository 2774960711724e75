//! Fused operator chains (one pass over HBM): `(x o1 y) o2 (z o3 w)` or `(x o1 y) o2 z`.
//!
//! The reference evaluates `(&nir - &red) / (nir + red)` (src/gdal/rasterband.rs:148,178) eagerly, one f64
//! temporary per operator.  Every step of such a chain is already an f64 rounded once (src/value.rs:207), so
//! `ec_fused` produces the same bits from a single kernel.  Operands are buffers or scalars; buffers of one cell
//! type, or of two (e.g. a UInt16 and a Float32 band), run as one pass without temporaries.
use crate::device::stream;
use crate::error::must;
use crate::ffi::*;
use crate::{BufferOps, CellBuffer, CellType, CellValue};
#[cfg(feature = "masked")]
use crate::{Mask, MaskedCellBuffer};
use std::os::raw::c_void;

/// One operand of a fused chain.
pub enum Operand<'a> {
    Buffer(&'a CellBuffer),
    Scalar(CellValue),
}

impl<'a> From<&'a CellBuffer> for Operand<'a> {
    fn from(b: &'a CellBuffer) -> Self {
        Operand::Buffer(b)
    }
}
impl<'a> From<CellValue> for Operand<'a> {
    fn from(v: CellValue) -> Self {
        Operand::Scalar(v)
    }
}
impl<'a> Operand<'a> {
    /// `Operand::scalar(2.0)`, `Operand::scalar(3u8)` — any primitive a cell can hold.
    pub fn scalar<T: Into<CellValue>>(v: T) -> Self {
        Operand::Scalar(v.into())
    }
}

/// `(x o1 y) o2 z` when `tail` is `None`, `(x o1 y) o2 (z o3 w)` when it is `Some((o3, w))`.
/// `o1`/`o2`/`o3` are `ffi::EC_ADD` … `ffi::EC_DIV`.  The result length is the shortest buffer operand's
/// (zip truncation of every step, src/buffer.rs:327); an empty result is a `UInt8` buffer (src/buffer.rs:233-234).
pub fn expr(x: Operand, o1: ec_op, y: Operand, o2: ec_op, z: Operand, tail: Option<(ec_op, Operand)>) -> CellBuffer {
    let (o3, w) = match tail {
        Some((o3, w)) => (o3, Some(w)),
        None => (-1, None), // EC_OP_NONE
    };
    let ops: Vec<&Operand> = [Some(&x), Some(&y), Some(&z), w.as_ref()].into_iter().flatten().collect();
    let mut dt = [0u8; 4];
    let mut p: [*const c_void; 4] = [std::ptr::null(); 4];
    let mut sc = [CellValue::UInt8(0).to_ffi(); 4];
    let mut n = usize::MAX;
    for (k, o) in ops.iter().enumerate() {
        match o {
            Operand::Buffer(b) => {
                dt[k] = b.cell_type() as u8;
                p[k] = b.dev_ptr();
                n = n.min(b.len());
            }
            Operand::Scalar(v) => sc[k] = v.to_ffi(),
        }
    }
    assert!(n != usize::MAX, "at least one operand must be a buffer");
    if n == 0 {
        return CellBuffer::empty_u8();
    }
    let out = CellBuffer::uninit(CellType::Float64, n);
    must(unsafe { ec_fused(o1, o2, o3, dt.as_ptr(), p.as_ptr(), sc.as_ptr(), n, out.mem.ptr() as *mut f64, stream()) }, "ec_fused");
    out
}

/// `(nir - red) / (nir + red)` in one pass.
pub fn ndvi(nir: &CellBuffer, red: &CellBuffer) -> CellBuffer {
    expr(nir.into(), EC_SUB, red.into(), EC_DIV, nir.into(), Some((EC_ADD, red.into())))
}

/// The masked form of [`ndvi`]: values as above over all cells, mask = AND of the operand masks
/// (src/masked/masked_buffer.rs:326-335 applied per step).
#[cfg(feature = "masked")]
pub fn ndvi_masked(nir: &MaskedCellBuffer, red: &MaskedCellBuffer) -> MaskedCellBuffer {
    let n = nir.len().min(red.len());
    if n == 0 {
        return MaskedCellBuffer::new(CellBuffer::empty_u8(), Mask::uninit(0));
    }
    let dt = [nir.cell_type() as u8, red.cell_type() as u8, nir.cell_type() as u8, red.cell_type() as u8];
    let p = [nir.buffer().dev_ptr(), red.buffer().dev_ptr(), nir.buffer().dev_ptr(), red.buffer().dev_ptr()];
    let m = [nir.mask().dev_ptr(), red.mask().dev_ptr(), nir.mask().dev_ptr(), red.mask().dev_ptr()];
    let (out, om) = (CellBuffer::uninit(CellType::Float64, n), Mask::uninit(n));
    must(
        unsafe {
            ec_masked_fused(EC_SUB, EC_DIV, EC_ADD, dt.as_ptr(), p.as_ptr(), m.as_ptr(), std::ptr::null(), n,
                            out.mem.ptr() as *mut f64, om.dev_ptr_mut(), stream())
        },
        "ec_masked_fused",
    );
    MaskedCellBuffer::new(out, om)
}
