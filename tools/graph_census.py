"""Node census of the captured hipGraphs of one train step: {kernel, memcpy, memset, ...} per capture.

    python tools/graph_census.py [--dtype bf16] [--small]

Why it exists: a memset node inside a captured region is a replay hazard on ROCm 7.2 (tools/graph_memset_probe.py,
models/decoder_graphs.py) -- the captured regions of this package are held to zero of them."""
import argparse
import os
import sys

os.environ["MEMOTR_GRAPH_CENSUS"] = "1"
os.environ.setdefault("MEMOTR_REQUIRE_GRAPHS", "1")
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32", choices=("f32", "bf16"))
    ap.add_argument("--encode-graphs", default="1")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    os.environ["MEMOTR_ENCODE_GRAPHS"] = args.encode_graphs
    from memotr_amd import train_bench
    from memotr_amd.models import decoder_graphs
    res = train_bench.run_train(args, 0, 1, dtype=args.dtype)
    print({k: res[k] for k in ("value", "ms_per_step", "decoder_graph_stats")})
    total = {}
    for i, c in enumerate(decoder_graphs.CENSUS):
        print(f"capture {i:3d}: {c}")
        for k, v in c.items():
            total[k] = total.get(k, 0) + v
    print("total:", total)


if __name__ == "__main__":
    main()
