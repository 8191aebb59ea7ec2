from .dataset import CropDataset, PredictionDataset, collate_fn
from .decoders import Decoder, FusedOutputDecoder, RawDecoder
from .transforms import Encode
from .augment import TrainAugmentation, ValidationAugmentation, pil_bilinear_coeffs, preprocess_images
from .feeder import BatchFeeder, GroupedBatch
