"""MI355X-native BigGAN training step: a drop-in for the generator/discriminator hot path of
david-jk/BigGAN-Tensorflow (ops.py, DiffAugment_tf.py, BigGAN.build_model train ops).

Python host code on PyTorch-ROCm tensors (device memory, streams, torch.distributed) calling the
hand-written gfx950 kernels of ``libbiggan_hip.so`` through the C ABI in ``include/biggan_hip.h``.
There is no CPU or PyTorch-eager fallback: every op raises if the library or a GPU is missing.
"""
__version__ = "0.1.0"
