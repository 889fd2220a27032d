"""Import alias: the package directory is `jpeg-decoder_amd/` (not a valid identifier), so
`import jpeg_decoder_amd` loads it through importlib and stands in for it."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("jpeg-decoder_amd")
