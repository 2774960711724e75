//! The reference's README example, on the GPU: `use erased_cells::…` resolves to this crate's library target.
//! (Uncompiled in this repository: no Rust toolchain in the build image.)
use erased_cells::{init, BufferOps, CellBuffer, CellType};

fn main() -> Result<(), Box<dyn std::error::Error>> {
    init(0)?;
    let cells = CellBuffer::from(vec![1u8, 2, 3]);
    let divisors = CellBuffer::from(vec![2u16, 4, 6]);
    // u8 / u16 -> Float64 on the device (every binary op widens both sides to f64), then a scalar RHS
    let result = cells / divisors * 0.5;
    assert_eq!(result.cell_type(), CellType::Float64);
    assert_eq!(result, vec![0.25, 0.25, 0.25].into());
    assert_eq!(result.to_vec::<f64>()?, vec![0.25; 3]);
    Ok(())
}
