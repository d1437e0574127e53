from .jyutvoice_tts import JyutVoiceTTS
from .text_encoder import TextEncoder

__all__ = ["TextEncoder", "JyutVoiceTTS"]
