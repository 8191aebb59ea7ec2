from .dataset import CropDataset, collate_fn
from .decoders import Decoder
from .transforms import Encode
