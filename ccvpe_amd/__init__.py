"""ccvpe_amd — the CCVPE dense cross-view matching path (inference + training step) on MI355X (gfx950).

    models      CVM_VIGOR / CVM_VIGOR_ori_prior / CVM_KITTI / CVM_OxfordRobotCar: the reference's module API and checkpoint
                layout; eval forward (fp32 / bf16), train mode through train.py
    train       train-mode forward with a tape + the explicit backward behind one torch.autograd.Function
    losses      infoNCELoss / cross_entropy_loss / orientation_loss (HIP forward + backward)
    targets     training ground truth built on the device;  optim: one-launch Adam;  preprocess: PIL-exact input transform
    ops, backward   one Python wrapper per C entry point of libccvpe_hip.so (include/ccvpe_hip.h), _lib: the ctypes table
    harness     replica timing harness, data-parallel gradient all-reduce (RCCL);  graph: hipGraph capture;  evaluate: sharded eval
    datasets    VIGOR / KITTI / Oxford RobotCar: split files -> index, decode, device batches;  repack: train-mode weight re-pack in one launch
    plan        the eval forward as one C call: record / serialise a launch plan, PlannedForward (ccvpe_ctx_create / ccvpe_forward)
    synth       deterministic synthetic weights / inputs shared by tests, goldens and bench.py

Everything arithmetic runs in libccvpe_hip.so (ccvpe_amd/csrc/*.hip); there is no CPU or eager fallback.
"""
