//! One process driving the GPUs of a node: a UInt16 raster cut into row-blocks, `raster / divisor` on every
//! shard, `min_max` of the whole raster through one all-reduce over xGMI.  (Uncompiled in this repository:
//! no Rust toolchain in the build image; the C twin `examples/sharded.c` runs in the GPU test suite.)
//!
//!   EC_HIP_LIB_DIR=../.. cargo run --example sharded_min_max -- 0 1 2 3 4 5 6 7
use erased_cells::ffi::EC_DIV;
use erased_cells::sharded::ShardGroup;
use erased_cells::CellValue;

fn main() -> Result<(), Box<dyn std::error::Error>> {
    let devices: Vec<i32> = std::env::args().skip(1).map(|a| a.parse().expect("device index")).collect();
    let devices = if devices.is_empty() { vec![0] } else { devices };
    let (rows, cols) = (4096u64, 4096u64);
    let raster: Vec<u16> = (0..rows * cols).map(|i| (1 + (i * 2654435761) % 65534) as u16).collect();
    let divisor: Vec<u16> = (0..rows * cols).map(|i| (1 + (i * 40503) % 65535) as u16).collect();

    let group = ShardGroup::new(&devices)?;
    let x = group.scatter(&raster, rows, cols)?;
    let d = group.scatter(&divisor, rows, cols)?;
    let q = x.binop(EC_DIV, &d)?; // Float64 on every shard, no communication
    let (mn, mx) = x.min_max()?; // per-shard keys, all-reduce(MAX) of 16 bytes, decode
    let (qmn, qmx) = q.min_max()?;
    assert_eq!((mn, mx), (CellValue::UInt16(*raster.iter().min().unwrap()), CellValue::UInt16(*raster.iter().max().unwrap())));
    println!("{} shard(s): raster min {mn:?} max {mx:?}; quotient min {qmn:?} max {qmx:?}", group.len());
    Ok(())
}
