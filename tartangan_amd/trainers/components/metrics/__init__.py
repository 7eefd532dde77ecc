from .fid import FIDComponent

__all__ = ['FIDComponent']
