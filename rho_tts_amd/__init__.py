"""MI355X-native generation provider for rho-tts (hand-written HIP for gfx950 behind a C ABI)."""
__version__ = "0.1.0"
