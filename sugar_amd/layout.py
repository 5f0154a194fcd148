"""
State-dict layouts of the reference DynamicTDNN supernet and of its cloned subnets.

The key names are the checkpoint format the inference path consumes
(``*.torchparams`` supernet + ``*.bn.tar`` subnet files, sugar/models/__init__.py:129-165):
  supernet keys : as registered by DynamicTDNN.__init__ (sugar/models/dynamictdnn.py:628-638) and the
                  Dynamic* primitives (sugar/modules/dynamicarch.py:100-113, 243-252, 367-399), e.g.
                  ``layer1.conv.k53``, ``layer1.conv.conv.weight``, ``layer1.bn.bn.running_mean``
  subnet keys   : as produced by DynamicTDNN.clone (dynamictdnn.py:713-730), one level shorter, e.g.
                  ``layer1.conv.weight``, ``blocks.0.resconv.convs.3.bn.weight``, ``bnpool.running_var``
"""
from collections import OrderedDict

SCALE = 8          # Res2 scale (dynamictdnn.py:384)
SE_REDUCTION = 4   # dynamictdnn.py:82
BOTTLENECK = 128   # dynamictdnn.py:494

_BN_LEAVES = (("weight", None), ("bias", None), ("running_mean", None), ("running_var", None),
              ("num_batches_tracked", ()))


def _as_list(v, n):
    return [v] * n if isinstance(v, int) else list(v)


def supernet_shapes(in_feats=80, out_embeds=192, num_blocks=4, channels=512, kernel_sizes=5,
                    catconv_channels=1536) -> "OrderedDict[str, tuple]":
    """{key: shape} in the reference's state_dict order for DynamicTDNN(...) (tdnn8m2g defaults)."""
    ch = _as_list(channels, num_blocks + 1)
    ks = _as_list(kernel_sizes, num_blocks + 1)
    d = OrderedDict()

    def dyn_conv(p, cin, cout, k):
        if k > 5:
            d[p + ".k75"] = (5, 5)
        if k > 3:
            d[p + ".k53"] = (3, 3)
        if k > 1:
            d[p + ".k31"] = (1, 1)
        d[p + ".conv.weight"] = (cout, cin, k)

    def dyn_bn(p, c):
        for leaf, shp in _BN_LEAVES:
            d["%s.bn.%s" % (p, leaf)] = (c,) if shp is None else shp

    def dyn_linear(p, cin, cout):
        d[p + ".linear.weight"] = (cout, cin)
        d[p + ".linear.bias"] = (cout,)

    dyn_conv("layer1.conv", in_feats, ch[0], ks[0])
    dyn_bn("layer1.bn", ch[0])
    for i in range(num_blocks):
        p = "blocks.%d." % i
        mid, k = ch[i + 1], ks[i + 1]
        dyn_conv(p + "first_conv.conv", ch[0], mid, 1)
        dyn_bn(p + "first_conv.bn", mid)
        for j in range(SCALE - 1):
            q = p + "resconv.convs.%d." % j
            dyn_conv(q + "conv", mid // SCALE, mid // SCALE, k)
            dyn_bn(q + "bn", mid // SCALE)
        dyn_conv(p + "last_conv.conv", mid, ch[0], 1)
        dyn_bn(p + "last_conv.bn", ch[0])
        dyn_linear(p + "se.reduce", ch[0], ch[0] // SE_REDUCTION)
        dyn_linear(p + "se.expand", ch[0] // SE_REDUCTION, ch[0])
    dyn_conv("catconv", ch[0] * num_blocks, catconv_channels, 1)
    dyn_conv("pooling.linear1", catconv_channels, BOTTLENECK, 1)
    dyn_conv("pooling.linear2", BOTTLENECK, catconv_channels, 1)
    dyn_bn("bnpool", catconv_channels * 2)
    dyn_linear("linear", catconv_channels * 2, out_embeds)
    for leaf, shp in _BN_LEAVES:
        d["bnembed." + leaf] = (out_embeds,) if shp is None else shp
    return d


def subnet_shapes(config, in_feats=80, out_embeds=192) -> "OrderedDict[str, tuple]":
    """{key: shape} of ``supernet.clone(config).state_dict()``; config = (depth, channels, kernels, catconv)."""
    depth, ch, ks, ccat = config
    c0 = ch[0]
    d = OrderedDict()

    def bn(p, c):
        for leaf, shp in _BN_LEAVES:
            d["%s.%s" % (p, leaf)] = (c,) if shp is None else shp

    d["layer1.conv.weight"] = (c0, in_feats, ks[0])
    bn("layer1.bn", c0)
    for i in range(depth):
        p = "blocks.%d." % i
        w = ch[i + 1] // SCALE
        mid = w * SCALE
        d[p + "first_conv.conv.weight"] = (mid, c0, 1)
        bn(p + "first_conv.bn", mid)
        for j in range(SCALE - 1):
            q = p + "resconv.convs.%d." % j
            d[q + "conv.weight"] = (w, w, ks[i + 1])
            bn(q + "bn", w)
        d[p + "last_conv.conv.weight"] = (c0, mid, 1)
        bn(p + "last_conv.bn", c0)
        r = c0 // SE_REDUCTION
        d[p + "se.reduce.weight"] = (r, c0)
        d[p + "se.reduce.bias"] = (r,)
        d[p + "se.expand.weight"] = (c0, r)
        d[p + "se.expand.bias"] = (c0,)
    d["catconv.weight"] = (ccat, c0 * depth, 1)
    d["pooling.linear1.weight"] = (BOTTLENECK, ccat, 1)
    d["pooling.linear2.weight"] = (ccat, BOTTLENECK, 1)
    bn("bnpool", ccat * 2)
    d["linear.weight"] = (out_embeds, ccat * 2)
    d["linear.bias"] = (out_embeds,)
    bn("bnembed", out_embeds)
    return d


def config_from_name(name: str):
    """``ecapa-tdnn.3.512.512.512.512.5.3.3.3.1536`` -> (3, [512]*4, [5,3,3,3], 1536)."""
    parts = [int(p) for p in name.split(".") if p.isdigit()]
    depth = parts[0]
    ch = parts[1:2 + depth]
    ks = parts[2 + depth:3 + 2 * depth]
    return (depth, ch, ks, parts[3 + 2 * depth])


def config_name(config, prefix="ecapa-tdnn") -> str:
    depth, ch, ks, ccat = config
    return ".".join([prefix, str(depth)] + [str(c) for c in ch] + [str(k) for k in ks] + [str(ccat)])


BASE = (3, [512, 512, 512, 512], [5, 3, 3, 3], 1536)
LARGEST = (4, [512] * 5, [5] * 5, 1536)
SMALLEST = (2, [128] * 3, [1] * 3, 384)
