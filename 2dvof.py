#!/usr/bin/env python3
"""2dvof.py -- drop-in command line of the reference solver, on the MI355X HIP library.

Same flags as /root/reference/2dvof.py:11-17 (`-ic {1,2,3}`, `-s`), same banner (:95-99), same
per-100-step status line (:533), same `output/NNNNNN-f.png` naming and plot (:563-571), same
`output/` and `data/` directories (:500-501).  The Taichi GUI is replaced by a headless loop;
everything numerical happens in libvof2d_hip.so (one C-ABI call per reference kernel, or the
fused vof_step).  There is no CPU engine behind this command.

Extensions (`python 2dvof.py -h`; the reference hard-codes them at :9,:19-20,:521 and loops until 'q'):
    --nx/--ny, --dtype f32|f64, --coord-cast, --steps N, --dt, --jacobi-iters N,
    --gpus N          row strips over N GPUs of this node (one process per GPU, RCCL halo exchange)
    --verbs           the literal main loop :513-528, one kernel per call
    --jacobi-tol T    residual-terminated pressure solve (--jacobi-max, --jacobi-crit abs|rel)
    --vis K           what the reference GUI would display (:531-559), saved with -s
    --save-every N    data/NNNNNNNN.npz checkpoints;  --resume FILE continues from one

The program itself is taichi-2d-vof_amd/vof2d/cli.py.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "taichi-2d-vof_amd"))

if __name__ == '__main__':
    from vof2d.cli import main
    raise SystemExit(main(os.path.abspath(__file__)))
