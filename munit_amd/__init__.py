"""munit_amd -- MI355X-native (gfx950) implementation of the MUNIT AdaINGen + MsImageDis
training step behind the reference's MUNIT_Trainer API.  See DESIGN.md / INTEGRATION.md."""
from .networks import AdaINGen, AdaINGen_double, MsImageDis  # noqa: F401
from .trainer import MUNIT_Trainer  # noqa: F401
from .utils import get_config, get_scheduler, weights_init  # noqa: F401

__all__ = ["AdaINGen", "AdaINGen_double", "MsImageDis", "MUNIT_Trainer", "get_config", "get_scheduler",
           "weights_init"]
