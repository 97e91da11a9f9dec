from .UPerNet import UPerNet
from .Projector import Projector
from .HRNet import hrnet48, hrnet32, hrnet18, HRNet
from .Swin import SwinTransformer
