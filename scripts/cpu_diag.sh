#!/bin/bash
# Host-side facts of the GPU box that decide how long the CPU restatement legs of the tests / bench take.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "nproc $(nproc)  affinity $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
cat /sys/fs/cgroup/cpu.max 2>/dev/null
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket|NUMA node\(s\)" 
free -g | head -2
df -h /dev/shm /tmp | tail -2
g++ -O3 -march=native -fopenmp -std=c++17 oracle/cpu_ref.cpp -o /tmp/cpu_ref_diag
python - <<'PY'
import numpy as np, subprocess, json, math, time, os
exe='/tmp/cpu_ref_diag'
dims=(128,64,64); N=dims[0]*dims[1]*dims[2]
ls_=tuple(math.pi*d/32 for d in dims)
rng=np.random.default_rng(1)
u0=2*0.4*(rng.random(N)-0.5); u1=u0+1e-3*(rng.random(N)-0.5)
u0.tofile('/tmp/u0.bin'); u1.tofile('/tmp/u1.bin')
p0,ds,theta=0.1,-0.001,0.5; p1=p0+ds/150
for env in ({}, {'OMP_NUM_THREADS':'8'}, {'OMP_NUM_THREADS':'16','OMP_WAIT_POLICY':'passive'}, {'OMP_NUM_THREADS':'8','OMP_WAIT_POLICY':'passive'}, {'OMP_NUM_THREADS':'4'}):
    t=time.time()
    r=subprocess.run([exe,*map(str,dims),*map(repr,ls_),'0.1','1.2','1.0',repr(ds),repr(theta),'/tmp/u0.bin',repr(p0),'/tmp/u1.bin',repr(p1),'1'],capture_output=True,text=True,env=dict(os.environ,**env))
    j=json.loads(r.stdout.strip().splitlines()[-1])
    print(env, 'step_s', j['seconds_per_step'], 'setup', j['setup_seconds'], 'threads', j['threads'], 'wall', round(time.time()-t,1), flush=True)
PY
