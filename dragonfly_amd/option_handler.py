"""Option specs -> Namespace, the subset of dragonfly/utils/option_handler.py:24-84 the GP
fitter needs (same function names and semantics; no argparse command line layer)."""
from argparse import Namespace
from copy import deepcopy


def get_option_specs(name, required=False, default=None, help_str='', **kwargs):
  """ option_handler.py:24-36 """
  ret = {'name': name, 'required': required, 'default': default, 'help': help_str}
  for key, value in list(kwargs.items()):
    ret[key] = value
  return ret


def load_options(list_of_options, descr='Algorithm', reporter=None, cmd_line=False,
                 partial_options=None):
  """ option_handler.py:51-84: defaults overridden by partial_options (Namespace or dict). """
  # pylint: disable=unused-argument
  opts = {}
  for elem in list_of_options:
    opts[elem['name']] = deepcopy(elem['default'])
  if partial_options is not None:
    items = partial_options.items() if isinstance(partial_options, dict) else \
            vars(partial_options).items()
    for key, value in items:
      opts[key] = value
  return Namespace(**opts)
