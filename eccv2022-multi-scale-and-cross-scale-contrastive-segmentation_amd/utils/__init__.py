from .datasets_info import DATASETS_INFO, register_dataset, num_all_classes, ignore_class
from .distributed import (is_distributed, get_rank, get_world_size, barrier, reduce_tensor,
                          all_reduce_numpy, concat_all_gather)
from .logger import Logger, printlog, set_verbosity
from .metrics import t_get_confusion_matrix, t_get_pixel_accuracy, t_get_mean_iou, t_get_miou
