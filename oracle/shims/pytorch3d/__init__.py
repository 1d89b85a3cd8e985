"""Stand-in for pytorch3d 0.7.0 (+ the reference's README.md:26-33 depth patch).
See ../README.md. Restated from the published PyTorch3D 0.7.0 algorithms; NOT the real package."""
__version__ = "0.7.0-shim"
