"""SKITGModel: the multi-object variant -- SinSKITGModel plus style-code conditioning.

Reference: /root/reference/models/skitG_model.py.  Its flags (:44-350) and forward
(:1284-1336, style code tiled over the innermost map and concatenated, networks.py:1600-1623)
are mirrored; its published optimize_parameters is broken (argument mismatches at :625-632 /
:651-653, SURVEY.md finding 3), so the train step follows the consistent SinSKITGModel
schedule, as the survey prescribes.  The style code is produced upstream by a frozen CLIP
ViT-B/32 (`net_style`, :484-489) whose weights cannot exist offline: this class consumes a
ready 512-d `style_code` from the batch (the synthetic dataset emits a seeded unit vector).
"""
from vts.misc import str2bool

from .sinskitG_model import SinSKITGModel, add_model_flags

B = str2bool

STYLE_FLAGS = [
    ("use_style_code", B, False), ("style_code_mode", str, "concat", ["concat", "adain"]),
    ("style_code_mapping_mode", str, "tile", ["tile", "project"]), ("style_code_dim", int, 512),
    ("num_layer_style_code", int, 1), ("use_external_test_input", B, False),
    ("test_sketch_material", str, "BlackJeans"), ("test_style_material", str, "BlackJeans"),
]


class SKITGModel(SinSKITGModel):
    MODEL_NAME = "skitG"
    DATASET_MODE = "skit"
    DATAROOT = "./datasets/singleskit_BluePants_padded_1800_x1/"
    DATA_LEN = 100

    @staticmethod
    def add_extra_flags(parser):
        add_model_flags(parser, STYLE_FLAGS)
        parser.add_argument("--material_list", type=str, nargs="+", default=[])
        parser.set_defaults(use_style_code=True)

    def _style(self):
        if not self.opt.use_style_code:
            return None
        if self.style_code is None:
            raise RuntimeError("skitG with --use_style_code True needs batch['style_code'] ([N, %d])" % self.opt.style_code_dim)
        return self.style_code
