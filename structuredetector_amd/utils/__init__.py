from .args import Arguments
from .ops import (clamp_in_0_1, clamped_sigmoid, decode_peaks, gather, gaussian_2d, hypot, nms, topk,
                  transpose_and_gather)
from .types import Box, ImageAnnotation, Keypoint, Object
from .misc import (AverageMeter, clip_annotation, dict_grouping, files_with_extension, get_unique_color_map, hflip_annotation, mkdir_if_needed,
                   set_seed, vflip_annotation)
from .visualization import draw, draw_embeddings, draw_heatmaps, draw_keypoints, draw_kp_and_emb, un_normalize
