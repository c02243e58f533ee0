"""ResNetBcos: torchvision-topology ResNet whose classifier runs BEFORE global average pooling, so that the
B-cosified `fc` (a 1x1 BcosifyConv2d) sees spatial features (reference bcos/models/standard_models.py:36-54)."""
try:  # the reference subclasses torchvision's class; use it when available so user code sees the same type
    from torchvision.models import ResNet
    from torchvision.models.resnet import BasicBlock, Bottleneck
except Exception:  # torchvision absent (MI355X image): restated topology with identical names
    from ._tv_resnet import BasicBlock, Bottleneck, ResNet

__all__ = ["ResNetBcos", "MyResNet", "BasicBlock", "Bottleneck"]


class MyResNet(ResNet):
    """Unmodified ordering (pool, flatten, fc): the non-B-cos baseline of the reference (:7-24)."""


class ResNetBcos(ResNet):
    def _forward_impl(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.fc(x)               # 1x1 B-cos conv on [N,C,h,w]
        return self.avgpool(x).flatten(1)
