from . import kaldi  # noqa: F401
