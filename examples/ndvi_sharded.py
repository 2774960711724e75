#!/usr/bin/env python3
"""BASELINE config 5: multi-band GeoTIFF -> convert u16 -> f32 -> NDVI, row-sharded over the ranks.

The reference's GDAL test (src/gdal/rasterband.rs:166-191) on its own fixtures, one process per GPU:
each rank reads only its row-block of the red / NIR bands, builds the nodata masks, converts to f32,
evaluates (nir - red) / (nir + red) on its shard (no communication), and the global data/nodata
counts and min/max come from the two scalar all-reduces (RCCL over xGMI with the nccl backend).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/ndvi_sharded.py
  # rehearsal on a 1-GPU box: add  --backend gloo --single-device
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "erased-cells_amd", "python"))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this pool

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import erased_cells_hip as ec  # noqa: E402
from erased_cells_hip import fused, raster, sharded  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--single-device", action="store_true")
    ap.add_argument("--fused", action="store_true", help="single-pass NDVI kernel instead of the eager chain")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, **({"device_id": torch.device("cuda", dev)} if args.backend == "nccl" else {}))
    ec.init(dev)

    red_rb = raster.RasterBand.open(os.path.join(args.data, "L8-Elkton-VA-B4.tiff"))
    nir_rb = raster.RasterBand.open(os.path.join(args.data, "L8-Elkton-VA-B5-nd.tiff"))
    cols, rows = red_rb.size()
    off, ln = sharded.shard_range(rows, cols, rank, world)
    row0, nrows = off // cols, ln // cols
    red = red_rb.read_cells_masked_rows(row0, nrows).convert(ec.Float32)
    nir = nir_rb.read_cells_masked_rows(row0, nrows).convert(ec.Float32)
    ndvi = fused.ndvi(nir, red) if args.fused else (nir - red) / (nir + red)

    data, nodata = sharded.sharded_counts(ndvi.mask())
    mn, mx = sharded.sharded_min_max(ndvi)
    if rank == 0:
        print(f"ranks {world}: rows/rank {nrows}  data {data} nodata {nodata}  NDVI min {float(mn.value)!r} max {float(mx.value)!r}")
        assert (data, nodata) == (31430, 4)
        assert float(mn.value).hex() == "-0x1.ff8ca5bcc77dcp-4" and float(mx.value).hex() == "0x1.5708125b0ed28p-1"
        print("matches the reference's known answers (src/gdal/rasterband.rs:150-160,180-188)")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
