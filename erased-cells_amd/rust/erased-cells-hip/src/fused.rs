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

/// One step of an expression program: `reg[dst] = a op b`.  Operands are made with [`stream`](Step::stream),
/// [`reg`](Step::reg), [`scalar`](Step::scalar).
pub type Step = ec_expr_step;

impl Step {
    pub const fn new(op: ec_op, a: i8, b: i8, dst: i8) -> Self {
        Step { op: op as i8, a, b, dst }
    }
    /// Operand: buffer `k` (0..=3) of the call.
    pub const fn stream(k: i8) -> i8 {
        ec_expr_stream(k)
    }
    /// Operand: register `k` (0..=3), written by an earlier step.
    pub const fn reg(k: i8) -> i8 {
        ec_expr_reg(k)
    }
    /// Operand: scalar `k` (0..=7) of the call.
    pub const fn scalar(k: i8) -> i8 {
        ec_expr_scalar(k)
    }
}

/// An operator tree of any depth in ONE pass (`ec_expr`): up to four buffers of any cell types, up to eight scalars,
/// up to sixteen steps over four f64 registers; the value is what the last step computed.  Same bits as the reference's
/// eager evaluation of the same operators in the same order (every intermediate is an f64 rounded once, src/value.rs:207).
/// EVI, `2.5 * (nir - red) / (nir + 6 * red - 7.5 * blue + 1)`:
/// ```ignore
/// use erased_cells::fused::{program, Step as S};
/// use erased_cells::ffi::{EC_ADD, EC_DIV, EC_MUL, EC_SUB};
/// let evi = program(&[&nir, &red, &blue], &[2.5.into(), 6.0.into(), 7.5.into(), 1.0.into()], &[
///     S::new(EC_SUB, S::stream(0), S::stream(1), 0), S::new(EC_MUL, S::reg(0), S::scalar(0), 0),
///     S::new(EC_MUL, S::stream(1), S::scalar(1), 1), S::new(EC_ADD, S::stream(0), S::reg(1), 1),
///     S::new(EC_MUL, S::stream(2), S::scalar(2), 2), S::new(EC_SUB, S::reg(1), S::reg(2), 1),
///     S::new(EC_ADD, S::reg(1), S::scalar(3), 1), S::new(EC_DIV, S::reg(0), S::reg(1), 0)]);
/// ```
///
/// # Panics
/// On a malformed program (the library's `EC_ERR_ARG`: a bad reference, a register read before it is written, counts
/// out of range).
pub fn program(streams: &[&CellBuffer], scalars: &[CellValue], steps: &[Step]) -> CellBuffer {
    assert!(!streams.is_empty() && streams.len() <= EC_EXPR_MAX_STREAMS, "1..=4 buffers");
    let n = streams.iter().map(|b| b.len()).min().unwrap_or(0);
    if n == 0 {
        return CellBuffer::empty_u8();
    }
    let dt: Vec<u8> = streams.iter().map(|b| b.cell_type() as u8).collect();
    let p: Vec<*const c_void> = streams.iter().map(|b| b.dev_ptr()).collect();
    let sc: Vec<ec_value> = scalars.iter().map(|v| v.to_ffi()).collect();
    let out = CellBuffer::uninit(CellType::Float64, n);
    must(
        unsafe {
            ec_expr(dt.as_ptr(), p.as_ptr(), streams.len() as i32, sc.as_ptr(), sc.len() as i32, steps.as_ptr(), steps.len() as i32, n,
                    out.mem.ptr() as *mut f64, stream())
        },
        "ec_expr",
    );
    out
}

/// The masked form of [`program`]: values over all cells, mask = AND of the streams' masks (what the eager chain of
/// `impl $trt for &MaskedCellBuffer`, src/masked/masked_buffer.rs:326-364, leaves behind).
#[cfg(feature = "masked")]
pub fn program_masked(streams: &[&MaskedCellBuffer], scalars: &[CellValue], steps: &[Step]) -> MaskedCellBuffer {
    assert!(!streams.is_empty() && streams.len() <= EC_EXPR_MAX_STREAMS, "1..=4 buffers");
    let n = streams.iter().map(|b| b.len()).min().unwrap_or(0);
    if n == 0 {
        return MaskedCellBuffer::new(CellBuffer::empty_u8(), Mask::uninit(0));
    }
    let dt: Vec<u8> = streams.iter().map(|b| b.cell_type() as u8).collect();
    let p: Vec<*const c_void> = streams.iter().map(|b| b.buffer().dev_ptr()).collect();
    let m: Vec<*const u8> = streams.iter().map(|b| b.mask().dev_ptr()).collect();
    let sc: Vec<ec_value> = scalars.iter().map(|v| v.to_ffi()).collect();
    let (out, om) = (CellBuffer::uninit(CellType::Float64, n), Mask::uninit(n));
    must(
        unsafe {
            ec_masked_expr(dt.as_ptr(), p.as_ptr(), m.as_ptr(), streams.len() as i32, sc.as_ptr(), sc.len() as i32, steps.as_ptr(),
                           steps.len() as i32, n, out.mem.ptr() as *mut f64, om.dev_ptr_mut(), stream())
        },
        "ec_masked_expr",
    );
    MaskedCellBuffer::new(out, om)
}

/// A host slice of any cell type as an operand of [`program_host`].
pub struct HostCells<'a> {
    ct: CellType,
    ptr: *const c_void,
    len: usize,
    nodata: Option<CellValue>,
    _borrow: std::marker::PhantomData<&'a ()>,
}

impl<'a> HostCells<'a> {
    pub fn new<T: crate::CellEncoding>(data: &'a [T]) -> Self {
        HostCells { ct: T::cell_type(), ptr: data.as_ptr() as *const c_void, len: data.len(), nodata: None, _borrow: std::marker::PhantomData }
    }
    /// The same with a nodata value (`from_vec_with_nodata`): cells equal to it are not valid.
    pub fn with_nodata<T: crate::CellEncoding>(data: &'a [T], nodata: T) -> Self {
        let mut s = Self::new(data);
        s.nodata = Some(CellValue::new(nodata));
        s
    }
}

/// Host memory in, host memory out (`ec_host_expr`): the program over host slices, streamed through the GPU in chunks with
/// upload, kernel and download overlapped — what the reference's `Vec`-in / `Vec`-out operators cost when nothing stays
/// resident (PCIe-bound: ≈ 6 Gcells/s at 16384², against ≈ 1.2 for `from_vec` + operator + `to_vec`).
pub fn program_host(streams: &[HostCells], scalars: &[CellValue], steps: &[Step]) -> Vec<f64> {
    assert!(!streams.is_empty() && streams.len() <= EC_EXPR_MAX_STREAMS, "1..=4 operands");
    let n = streams.iter().map(|s| s.len).min().unwrap_or(0);
    let dt: Vec<u8> = streams.iter().map(|s| s.ct as u8).collect();
    let p: Vec<*const c_void> = streams.iter().map(|s| s.ptr).collect();
    let sc: Vec<ec_value> = scalars.iter().map(|v| v.to_ffi()).collect();
    let mut out = vec![0f64; n];
    must(
        unsafe {
            ec_host_expr(dt.as_ptr(), p.as_ptr(), streams.len() as i32, sc.as_ptr(), sc.len() as i32, steps.as_ptr(), steps.len() as i32, n,
                         out.as_mut_ptr(), 0)
        },
        "ec_host_expr",
    );
    out
}

/// The masked form (`ec_host_masked_expr`): `from_vec_with_nodata` of every operand made with [`HostCells::with_nodata`], the
/// program with the AND of the masks, `to_vec_with_nodata(out_nodata)` of the f64 result — in one streamed call.  Returns the
/// values and the validity of every cell.
pub fn program_host_masked(streams: &[HostCells], scalars: &[CellValue], steps: &[Step], out_nodata: Option<f64>) -> (Vec<f64>, Vec<bool>) {
    assert!(!streams.is_empty() && streams.len() <= EC_EXPR_MAX_STREAMS, "1..=4 operands");
    let n = streams.iter().map(|s| s.len).min().unwrap_or(0);
    let dt: Vec<u8> = streams.iter().map(|s| s.ct as u8).collect();
    let p: Vec<*const c_void> = streams.iter().map(|s| s.ptr).collect();
    let nd_values: Vec<Option<ec_value>> = streams.iter().map(|s| s.nodata.map(|v| v.to_ffi())).collect();
    let nd: Vec<*const ec_value> = nd_values.iter().map(|v| v.as_ref().map_or(std::ptr::null(), |x| x as *const ec_value)).collect();
    let sc: Vec<ec_value> = scalars.iter().map(|v| v.to_ffi()).collect();
    let mut out = vec![0f64; n];
    let mut mask = vec![0u8; n];
    let ond = out_nodata.unwrap_or(0.0);
    must(
        unsafe {
            ec_host_masked_expr(dt.as_ptr(), p.as_ptr(), nd.as_ptr(), streams.len() as i32, sc.as_ptr(), sc.len() as i32, steps.as_ptr(),
                                steps.len() as i32, n, out.as_mut_ptr(), if out_nodata.is_some() { &ond } else { std::ptr::null() },
                                mask.as_mut_ptr(), 0)
        },
        "ec_host_masked_expr",
    );
    (out, mask.into_iter().map(|b| b != 0).collect())
}

/// `(min, max)` of a program's result without its raster (`ec_expr_min_max`): once the library has compiled the program for
/// itself only the streams are read; until then it runs the program into a temporary and reduces that.  Both values are
/// `Float64`; an empty input gives the fold's identities `(f64::MAX, f64::MIN)`, like `BufferOps::min_max`.
pub fn program_min_max(streams: &[&CellBuffer], scalars: &[CellValue], steps: &[Step]) -> (CellValue, CellValue) {
    assert!(!streams.is_empty() && streams.len() <= EC_EXPR_MAX_STREAMS, "1..=4 buffers");
    let n = streams.iter().map(|b| b.len()).min().unwrap_or(0);
    let dt: Vec<u8> = streams.iter().map(|b| b.cell_type() as u8).collect();
    let p: Vec<*const c_void> = streams.iter().map(|b| b.dev_ptr()).collect();
    let sc: Vec<ec_value> = scalars.iter().map(|v| v.to_ffi()).collect();
    let (mut mn, mut mx) = (CellValue::Float64(0.0).to_ffi(), CellValue::Float64(0.0).to_ffi());
    must(
        unsafe {
            ec_expr_min_max(dt.as_ptr(), p.as_ptr(), std::ptr::null(), streams.len() as i32, sc.as_ptr(), sc.len() as i32, steps.as_ptr(),
                            steps.len() as i32, n, &mut mn, &mut mx, stream())
        },
        "ec_expr_min_max",
    );
    (CellValue::from_ffi(&mn), CellValue::from_ffi(&mx))
}
