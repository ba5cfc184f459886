"""`from models.stylegan2.model import Generator` keeps working: the MI355X-native generator
(same constructor, forward signature and state_dict keys as models/stylegan2/model.py:457-648)."""
from ganecdotes_amd.generator import Generator    # noqa: F401
