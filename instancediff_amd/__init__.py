"""MI355X-native hot path of InstanceDiff (score-SDE UNet + iterative denoising loop).

Python host code (reference option / plugin surface) over a C ABI of hand-written gfx950 HIP kernels
(include/idiff.h, instancediff_amd/csrc/).  See DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"
