"""Package logger (reference: chroma/log.py:1-3)."""
import logging

logger = logging.getLogger("chroma")
if not logger.handlers:
    logger.addHandler(logging.NullHandler())
