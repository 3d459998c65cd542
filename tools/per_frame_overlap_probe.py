"""Per-frame stage pipeline (bench_dropin.dropin_surface, the reference's structure: one frame per stage call,
validate='always', 8 frames in flight) under the measurement switches that let the encoder stage's launches and the
recon stage's query launches share the chip:
  MONOPORT_QUERY_WGS_PER_CU=1      the persistent query kernel leaves one workgroup slot per CU free
  MONOPORT_QUERY_GRID_MULT=M       the query launches M x the resident workgroups (slots come back a few at a time)
  MONOPORT_STAGE_PRIORITY=4:-1     HIP priority of a stage's stream (4 = netG.filter, 5 = reconEngine)
  MONOPORT_PLAN_SIDE_PRIORITY=-1   priority of the recorded encoder plan's side streams
Prints one line: recon/s (median / min / max of the passes) and the single-frame latency."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_dropin  # noqa: E402

if __name__ == "__main__":
    label = sys.argv[1] if len(sys.argv) > 1 else "run"
    res = bench_dropin.dropin_surface(torch.device("cuda:0"), 96, 3, [17, 33, 65, 129, 257], passes=5, legs=("per_frame",))
    pf = res["per_frame_stages"]
    sw = {k: v for k, v in os.environ.items() if k.startswith("MONOPORT_")}
    print("[%s] per-frame stages %.1f recon/s (min %.1f max %.1f), single-frame latency %.2f ms  %s"
          % (label, pf["value"], pf["passes"]["value_min"], pf["passes"]["value_max"], res["latency_ms_single_frame"], sw), flush=True)
