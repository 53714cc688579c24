"""Named model / hyper-parameter sets of the reference's yaml tree, as Python data (for bench.py, smoke and tests that
must not read /root/reference).  Values: apps/mobilenet/models/fix_first/supernet_single_path_nas*.yml,
apps/mobilenet/models/mobilenet_v2_1.0*.yml, apps/slimming/shrink/atomnas_{a,c}.yml, apps/mobilenet/default_mnas_scheduler.yml."""

_SUPERNET_ROWS = [[1, 16, 1, 1, [3]], [6, 24, 4, 2, [3, 5, 7]], [6, 40, 4, 2, [3, 5, 7]], [6, 80, 4, 2, [3, 5, 7]],
                  [6, 96, 4, 1, [3, 5, 7]], [6, 192, 4, 2, [3, 5, 7]], [6, 320, 1, 1, [3, 5, 7]]]
_MBV2_ROWS = [[1, 16, 1, 1, [3]], [6, 24, 2, 2, [3]], [6, 32, 3, 2, [3]], [6, 64, 4, 2, [3]], [6, 96, 3, 1, [3]],
              [6, 160, 3, 2, [3]], [6, 320, 1, 1, [3]]]


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
