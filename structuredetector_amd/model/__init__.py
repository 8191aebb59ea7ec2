from .evaluator import Evaluation, Evaluations, Evaluator
from .loss import Loss, LossStats
from .network import Network
from .export import FusedInferenceModel
