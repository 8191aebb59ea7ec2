from .dataset import collate_fn
from .decoders import Decoder
from .transforms import Encode
