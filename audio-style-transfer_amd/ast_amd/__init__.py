"""ast_amd -- MI355X-native (gfx950) hot path of francescobrigante/Audio-Style-Transfer.

Drop-in modules keep the reference's call surface (StyleEncoder, ContentEncoder,
Decoder, Discriminator, losses, utilityFunctions); all arithmetic runs in
hand-written HIP kernels behind the C-ABI of include/ast_hip.h.  There is no CPU
or ATen fallback: without libast_hip.so and a GPU the ops raise.
"""
from . import config  # noqa: F401
from .config import set_compute_dtype  # noqa: F401
from .style_encoder import StyleEncoder, SinusoidalPositionalEncoding, initialize_weights  # noqa: F401
from .content_encoder import ContentEncoder  # noqa: F401
from .new_decoder import Decoder, compute_comprehensive_loss  # noqa: F401
from .discriminator import Discriminator  # noqa: F401
from .losses import infoNCE_loss, margin_loss, adversarial_loss, disentanglement_loss  # noqa: F401
