"""hydra-mi: MI355X-native hot path of HydraGL (hydradarpa/kalman-hydra).

Brox optical flow followed by the EKF measurement update of a textured triangle
mesh, as hand-written gfx950 HIP kernels behind a C-ABI (include/hydra_mi.h),
with the reference's Python interface on top:

    brox      BroxOpticalFlow            (replaces cv::cuda::BroxOpticalFlow, src/optical_flow_ext.cpp)
    renderer  Renderer, FlowStream       (replaces renderer.py + cuda.py / cuda_multi.py)
    kalman    KalmanFilter, IteratedKalmanFilter, IteratedMSKalmanFilter, stats   (kalman.py)
    matio     the .mat flow file format  (src/optical_flow_ext.cpp:47-170)
    synth     synthetic inputs           (synth.py, synthetic/flowfields.py)

The directory name is fixed by the build contract; ``import hydra_mi`` is the
importable alias.
"""
__version__ = "0.1"
