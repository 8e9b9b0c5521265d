from . import util, filters  # placeholders: imported by the reference's vocoder utils, never called
