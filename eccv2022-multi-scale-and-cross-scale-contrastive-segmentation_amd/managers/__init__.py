from .BaseManager import BaseManager
from .HRNet_Manager import HRNetManager
from .OCRNet_Manager import OCRNetManager
