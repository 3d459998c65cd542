"""Time the REFERENCE's own CPU path (its Python modules imported from /root/reference) in the
build container: BASELINE.json configs[0] (dense 64^3 through netG.query) and one 256^3
reconstruction (netG.filter + our octree schedule driving the reference's netG.query + the
reference's forward_vertices).  Prints one JSON object; the numbers are recorded in BASELINE.md as
cpu baseline kind "reference" (the GPU box has no /root/reference, so bench.py times the port).

    python oracle/time_reference.py [threads]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("MONOPORT_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(HERE, "refshim"), REF, os.path.join(REF, "RTL"), ROOT]

import torch  # noqa: E402

from monoport_amd import synthetic as syn  # noqa: E402
from oracle import pifu_oracle as orc  # noqa: E402
from oracle.gen_golden import dense_lattice, load_mlp, ref_net  # noqa: E402


def median(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), out


@torch.no_grad()
def main():
    import recon as ref_recon
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    torch.set_num_threads(threads)
    net = ref_net("G")
    shapes = {k: tuple(v.shape) for k, v in net.image_filter.state_dict().items()}
    net.image_filter.load_state_dict(
        {k: torch.from_numpy(v) for k, v in syn.seeded_state_dict(shapes, 71).items()})
    layers = syn.body_mlp("G", noise=0.05, seed=1)
    load_mlp(net, layers)
    img = torch.from_numpy(syn.synthetic_image(0))[None]
    calib = ref_recon.pifu_calib(*syn.scene_camera(0), device="cpu")
    planes = torch.from_numpy(syn.body_feature_planes(128, 128))

    net.filter(img)  # warm-up
    t_filter, feats = median(lambda: net.filter(img))
    feats[-1][0][0, 0:2].copy_(planes)
    p64 = torch.from_numpy(dense_lattice(64))[None]
    net.query(feats, p64[:, :, :4096], calibs=calib)
    t_dense, _ = median(lambda: net.query(feats, p64, calibs=calib)[0])

    def query_func(points):  # RTL/main.py:169-183 on [3,N] numpy
        pt = torch.from_numpy(points.T.copy())[None]
        samples = pt.repeat(1, 1, 1).permute(0, 2, 1)
        return net.query(feats, points=samples, calibs=calib)[0][0, 0].numpy()

    stats = []

    def octree():
        stats.clear()
        return orc.seg3d_lossless(query_func, [-1, -1, -1], [1, 1, 1], [17, 33, 65, 129, 257], stats=stats)

    t_oct, sdf = median(octree, 3)
    t_fv, _ = median(lambda: ref_recon.forward_vertices(torch.from_numpy(sdf)[None, None], "front"))
    total = t_filter + t_oct + t_fv
    print(json.dumps({
        "kind": "reference", "where": "build container (no GPU)", "threads": threads,
        "netG.filter_s": t_filter, "netG.query_dense64_s": t_dense,
        "dense64_mpts_per_s": 64 ** 3 / t_dense / 1e6,
        "octree257_reference_query_s": t_oct, "octree_points": int(sum(stats)),
        "octree_mpts_per_s": sum(stats) / t_oct / 1e6, "forward_vertices_s": t_fv,
        "recon_per_s": 1.0 / total, "s_per_recon": total}))


if __name__ == "__main__":
    main()
