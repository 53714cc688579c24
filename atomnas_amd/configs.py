"""Named model / hyper-parameter sets of the reference's yaml tree, as Python data (for bench.py, smoke and tests that
must not read /root/reference).  Values: apps/mobilenet/models/fix_first/supernet_single_path_nas*.yml,
apps/mobilenet/models/mobilenet_v2_1.0*.yml, apps/slimming/shrink/atomnas_{a,c}.yml, apps/mobilenet/default_mnas_scheduler.yml."""

_SUPERNET_ROWS = [[1, 16, 1, 1, [3]], [6, 24, 4, 2, [3, 5, 7]], [6, 40, 4, 2, [3, 5, 7]], [6, 80, 4, 2, [3, 5, 7]],
                  [6, 96, 4, 1, [3, 5, 7]], [6, 192, 4, 2, [3, 5, 7]], [6, 320, 1, 1, [3, 5, 7]]]
_MBV2_ROWS = [[1, 16, 1, 1, [3]], [6, 24, 2, 2, [3]], [6, 32, 3, 2, [3]], [6, 64, 4, 2, [3]], [6, 96, 3, 1, [3]],
              [6, 160, 3, 2, [3]], [6, 320, 1, 1, [3]]]


# searched architectures (rows [c, n, s, ks, hiddens, expand]): apps/searched/models/atomnas_{a,c}.yml
_ATOMNAS_C_ROWS = [[16, 1, 1, [3], [32], False], [24, 1, 2, [3, 5, 7], [15, 23, 13], True], [24, 1, 1, [3, 5, 7], [20, 3, 2], True],
                   [24, 1, 1, [3, 5, 7], [21, 8, 1], True], [24, 1, 1, [3, 5], [21, 4], True], [40, 1, 2, [3, 5, 7], [64, 65, 70], True],
                   [40, 1, 1, [3, 5, 7], [40, 23, 28], True], [40, 1, 1, [3, 5, 7], [36, 21, 9], True], [40, 1, 1, [3, 5, 7], [40, 22, 7], True],
                   [80, 1, 2, [3, 5, 7], [169, 170, 157], True], [80, 1, 1, [3, 5, 7], [45, 55, 74], True],
                   [80, 1, 1, [3, 5, 7], [58, 21, 37], True], [80, 1, 1, [3, 5, 7], [77, 18, 39], True],
                   [96, 1, 1, [3, 5, 7], [393, 263, 247], True], [96, 1, 1, [3, 5, 7], [57, 40, 36], True],
                   [96, 1, 1, [3, 5, 7], [61, 34, 26], True], [96, 1, 1, [3, 5, 7], [87, 38, 33], True],
                   [192, 1, 2, [3, 5, 7], [454, 468, 468], True], [192, 1, 1, [3, 5, 7], [188, 137, 306], True],
                   [192, 1, 1, [3, 5, 7], [253, 187, 314], True], [192, 1, 1, [3, 5, 7], [345, 202, 233], True],
                   [320, 1, 1, [3, 5, 7], [823, 738, 749], True]]


def searched_kwparams(name):
    """kwargs of models.searched_network.Model.  'atomnas_c': apps/searched/atomnas_c/atomnas_c.yml (retrain of the searched net);
    'atomnas_c_plus': .../atomnas_c+.yml = BASELINE config 5 (SE ratio 0.5, Swish, fused blocks, dropout 0.28)."""
    base = dict(input_channel=32, last_channel=1280, width_mult=1.0, round_nearest=8, active_fn='nn.ReLU', num_classes=1000,
                inverted_residual_setting=[[r[0], r[1], r[2], list(r[3]), list(r[4]), r[5]] for r in _ATOMNAS_C_ROWS],
                dropout_ratio=0.28, batch_norm_momentum=0.01, batch_norm_epsilon=1e-3)
    if name == 'atomnas_c':
        return base
    if name == 'atomnas_c_plus':
        return dict(base, se_ratio=0.5, active_fn='nn.Swish', block='InvertedResidualChannelsFused')
    raise KeyError(name)


def model_kwparams(name):
    base = dict(active_fn='nn.ReLU', num_classes=1000, last_channel=1280, width_mult=1.0, round_nearest=8,
                batch_norm_momentum=0.01, batch_norm_epsilon=1e-3)
    if name == 'atomnas_c_supernet':
        return dict(base, input_channel=32, inverted_residual_setting=[list(r) for r in _SUPERNET_ROWS])
    if name == 'atomnas_a_supernet':
        return dict(base, input_channel=16, inverted_residual_setting=[list(r) for r in _SUPERNET_ROWS])
    if name == 'mobilenet_v2_1.0':
        return dict(base, input_channel=32, inverted_residual_setting=[list(r) for r in _MBV2_ROWS])
    raise KeyError(name)


# resolved search hyper-parameters ([probed] from the reference's config loader)
SEARCH_HPARAMS = dict(optimizer='rmsprop', momentum=0.9, alpha=0.9, epsilon=0.001, eps_inside_sqrt=True, weight_decay=1e-5,
                      weight_decay_method='mnas', base_lr=0.016, base_total_batch=256, lr_scheduler='exp_decaying',
                      lr_stepwise=False, exp_decaying_lr_gamma=0.97, exp_decay_epoch_interval=2.4, label_smoothing=0.1,
                      moving_average_decay=0.9999, moving_average_decay_adjust=True, moving_average_decay_base_batch=4096,
                      num_epochs=350, image_size=224, random_seed=1995,
                      prune_params=dict(method='network_slimming', bn_prune_filter='expansion_only_skip_expand1', rho=1e-4,
                                        epoch_free=0, epoch_warmup=25, scheduler='linear', stepwise=True, logging_verbose=False),
                      model_shrink_threshold=1e-3, model_shrink_delta_flops=1e6)
