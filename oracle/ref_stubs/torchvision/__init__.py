"""Placeholder so that `from torchvision import models` (modules/models/lseg_vit_zs.py:8, ResNet backbones only) imports.
TEST INFRASTRUCTURE."""
import types

models = types.ModuleType("torchvision.models")
