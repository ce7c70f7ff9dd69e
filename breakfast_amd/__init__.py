"""breakfast_amd: MI355X-native hot path (token vocabulary -> CSR -> all-pairs set-difference distance
within --max-dist -> connected components) behind the function-level API of rki-mf1/breakfast."""

__version__ = "0.1.0"
