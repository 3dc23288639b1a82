from .PixArtMS import PixArtMS, PixArtMS_XL_2, PixArtMSBlock  # noqa: F401
