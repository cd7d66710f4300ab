"""MI355X-native hot path of aangelopoulos/im2im-uq."""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: every launch's argument fetch stays on the device.  A training step
# is a chain of ~200 dependent launches, 100 of them a few microseconds long: per-GPU batch 10 (the 8-GPU share of the reference's batch
# of 78) 6.57 -> 6.39 ms, batch 78 38.23 -> 37.99 ms (three alternating pairs on one box, profiles/r06_ab_experiments.txt section 5).  The
# HIP runtime reads the variable when it initialises, i.e. at the process's first HIP call: import this package (or set the variable
# yourself) before touching the GPU.  An explicit setting in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
