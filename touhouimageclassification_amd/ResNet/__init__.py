"""ResNet path: ``model.resnet18 .. resnet152`` over the HIP conv / BatchNorm kernels, ``train`` (SGD harness)."""
