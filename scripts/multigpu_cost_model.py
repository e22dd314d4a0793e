"""Cost model of the 512^3 corrector step on 2 / 4 / 8 MI355X from MEASURED 1-GPU kernel times (no multi-GPU node was
available to this round; every RCCL number below is an assumption, stated).

Inputs (written by scripts/gpu_cost_model_inputs.sh on the GPU box, copied to profiles/):
  bench_slab{64,128,256}.json   bench.py --size 512 --size-z nz: the corrector step on the z-slab one of 8 / 4 / 2 ranks owns,
                                i.e. every local kernel at its distributed size with the real iteration counts
  slabemu.jsonl                 one preconditioner application on that slab: transposed-DCT local part (5 passes) vs the slab
                                z-solve's local part (6 passes + face kernels), kernel_sweep.py slabemu
  bench 512^3 1-GPU line        the denominator
Assumptions: xGMI 7 links x 45 GB/s effective per direction and GPU for the all-to-alls (the r1 figure), 25 us per RCCL
collective / send-recv group, halo exchange (4 MiB per face) hidden behind the interior z-chunks, one scalar all-reduce per
multidot / dot (batched)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
one = json.load(open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r3_bench_512_1gpu.json")))
emu = [json.loads(l) for l in open(os.path.join(src, "slabemu.jsonl"))]
LAT, LINK = 25e-6, 7 * 45e9
rows = []
for R, nz in ((2, 256), (4, 128), (8, 64)):
    b = json.load(open(os.path.join(src, f"bench_slab{nz}.json")))
    k = b["kernels"]
    steps = b["steps"]
    t_local = b["ms_per_step"] * 1e-3                                  # all local kernels + host gaps, z pass of length nz
    applies = k["dct_pass"]["calls"] / steps / 5.0
    e0 = next(e for e in emu if e["R"] == R and not e["slab_zsolve"])["ms"] * 1e-3
    # the slab z-solve in the form the library runs by default: forward / inverse halves where measured (round 5), else two round trips
    e1 = min(e["ms"] for e in emu if e["R"] == R and e["slab_zsolve"]) * 1e-3
    n_red = (k["multidot"]["calls"] + k["blas1"]["calls"] * 0.3) / steps      # multidots + the dots / norms among the BLAS-1 calls
    vec_bytes = 8.0 * 512 ** 3 / R
    t_a2a_T = 2 * applies * (LAT + vec_bytes * (R - 1) / R / LINK)      # two transposes of the slab per application
    t_a2a_S = 2 * applies * (LAT + 32.0 * 512 * 512 / LINK)             # 4 doubles per line each way
    t_red = n_red * LAT
    t_T = t_local + t_a2a_T + t_red
    t_S = t_local + applies * (e1 - e0) + t_a2a_S + t_red
    # two lanes (docs/history.md 8c): the two right-hand sides of the bordered solve in flight at once, each lane with its own
    # communicator -- measured local step with two lanes; ASSUMED: half of the collective time of one solve hides behind the
    # other solve's kernels (the second z round trip is real work and stays)
    two = None
    f2 = os.path.join(src, f"bench_slab{nz}_two_lanes.json")
    if os.path.exists(f2):
        b2 = json.load(open(f2))
        t_loc2 = b2["ms_per_step"] * 1e-3
        t_2 = t_loc2 + applies * (e1 - e0) + 0.5 * (t_a2a_S + t_red)
        two = dict(local_ms=t_loc2 * 1e3, hidden_fraction_of_collectives=0.5, step_ms=t_2 * 1e3,
                   speedup=one["ms_per_step"] * 1e-3 / t_2)
    rows.append(dict(gpus=R, slab_planes=nz, local_ms=t_local * 1e3, precond_applies=applies, two_lanes=two,
                     precond_local_ms=dict(transposed=e0 * 1e3, slab_zsolve=e1 * 1e3), allreduce_ms=t_red * 1e3,
                     transposed=dict(alltoall_ms=t_a2a_T * 1e3, step_ms=t_T * 1e3, speedup=one["ms_per_step"] * 1e-3 / t_T),
                     slab_zsolve=dict(alltoall_ms=t_a2a_S * 1e3, step_ms=t_S * 1e3, speedup=one["ms_per_step"] * 1e-3 / t_S),
                     ideal_speedup_local_only=one["ms_per_step"] / b["ms_per_step"]))
print(json.dumps(dict(one_gpu_ms_per_step=one["ms_per_step"], assumptions=dict(latency_us=25, alltoall_GBs_per_gpu=LINK / 1e9),
                      model=rows), indent=1))
