"""A/B builds of the product library for tuning runs: the same sources with extra compiler flags / -D switches, linked to
raytracingdenoiser_amd/lib/variants/<name>/libNRD_hip.so (objects under /tmp). bench.py and the tests pick one up through NRD_HIP_LIBRARY=<path>.
usage: python tools/build_variant.py NAME [extra hipcc flags ...]      e.g.  python tools/build_variant.py ilp -mllvm -amdgpu-sched-strategy=max-ilp
A variant that changes arithmetic must be held against the oracle like the product:  NRD_HIP_LIBRARY=<path> python -m pytest tests -m gpu"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracingdenoiser_amd import build as B  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    out = os.path.join(B.LIB_DIR, "variants", name, B.LIB_NAME)
    print(B.build_product(verbose=False, out=out, extra=extra, obj_dir=os.path.join("/tmp/nrd_variants", name)))


if __name__ == "__main__":
    main()
