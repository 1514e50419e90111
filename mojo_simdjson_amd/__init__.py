"""mojo-simdjson_amd: MI355X-native stage-1 JSON structural indexer.

One hot path of gabrieldemarmiesse/mojo-simdjson (stage 1, the structural
indexer) rebuilt as hand-written HIP for gfx950 behind a C ABI
(include/msj_stage1.h).  See DESIGN.md.
"""
from . import errors  # noqa: F401
from .dom_parser_implementation import DomParserImplementation  # noqa: F401
