"""Test flags (mirror of /root/reference/options/test_options.py:11-22)."""
from .base_options import FLAG, BaseOptions, add_table

TEST_TABLE = [
    ("phase", str, "test"),
    ("eval", FLAG, False),
    ("num_test", int, 50),
]


class TestOptions(BaseOptions):
    def initialize(self, parser):
        parser = BaseOptions.initialize(self, parser)
        add_table(parser, TEST_TABLE)
        parser.set_defaults(load_size=parser.get_default("crop_size"))
        self.isTrain = False
        return parser
