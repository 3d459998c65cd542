"""Minimal stand-in for yacs.config.CfgNode (attribute dict).

Test infrastructure only: lets oracle/gen_golden.py import the *reference*
MonoPort modules from /root/reference in the build container, where the real
`yacs` package is absent.  Never imported by the product.
"""


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v
