from .synthetic import SyntheticSegmentation
