#!/usr/bin/env python
"""AtomNAS supernet training on MI355X:  python train.py app:apps/slimming/shrink/atomnas_c.yml [--dotted.key value ...]

Same entry, yaml semantics and loop order as the reference's train.py (run_one_epoch :130-250, per-epoch validate / mask /
shrink :364-411, checkpoint :395-411), with the iteration executed by atomnas_amd.engine.TrainStep (one replayed hipGraph
per iteration, RCCL all-reduce of the gradient arena).  Under a launcher (torch.distributed.run) one process drives one GPU.
Input: `dataset: imagenet1k_fake` (synthetic, device resident: the benchmark protocol) or any decoded source through the reference's
factories utils.dataflow.data_transforms / dataset / data_loader (`dataset: imagenet1k_decoded_fake`, or a module of your own) ->
utils.dataflow.DevicePrefetcher (crop / PIL-exact bilinear resize / flip / normalize on the GPU) -> TrainStep.set_batch.  JPEG decoding
and LMDB reading are not available in this image.
"""
import logging
import os
import sys
import time

import torch

from atomnas_amd import engine
from atomnas_amd.models import mobilenet_base as mb
from atomnas_amd.utils import config as cfg
from atomnas_amd.utils import distributed as udist
from atomnas_amd.utils import optim, prune
from atomnas_amd.utils.common import bn_calibration, get_params_by_name, set_random_seed

NUM_IMAGENET_TRAIN = 1281167


LOADERS = None   # (train_loader, calib_loader, val_loader, test_loader) when the yaml names a decoded data source; None: synthetic batches


def device_batches(loader, steps):
    """at most `steps` batches of `loader` through the GPU input pipeline (train.py:215-218 of the reference: DataPrefetcher)"""
    from atomnas_amd.utils import dataflow
    tf = getattr(getattr(loader, 'dset', None), 'transform', None)   # the split's DeviceTransform: resampling filter, mean, std
    kw = dict(filter=tf.filter, mean=tf.mean, std=tf.std) if isinstance(tf, dataflow.DeviceTransform) else {}
    pre = dataflow.DevicePrefetcher(loader, image_size=cfg.FLAGS.image_size, **kw)
    try:
        for i, (x, y) in enumerate(pre):
            if i >= steps:
                break
            yield x, y
    finally:
        pre.close()


def fake_batches(batch, image_size, num_classes, steps, seed):
    """Device-resident synthetic batches (normal images, uniform labels); the reference's FakeData is all zeros."""
    g = torch.Generator(device='cuda').manual_seed(seed)
    x = torch.randn(batch, 3, image_size, image_size, device='cuda', generator=g)
    y = torch.randint(0, num_classes, (batch,), device='cuda', generator=g)
    for _ in range(steps):
        yield x, y


def shrink_model(model_wrapper, ema, optimizer, prune_info, threshold=1e-3, ema_only=False):
    """Discard dead atomic blocks (train.py:27-81): mask = |gamma| > thr OR |gamma_ema| > thr (EMA only at the last epoch)."""
    import common as mc
    FLAGS = cfg.FLAGS
    model = mc.unwrap_model(model_wrapper)
    # all alive masks in one launch over the parameter / EMA arenas (the reference: 63 x abs, compare, or, sum().item())
    from atomnas_amd import runtime
    blocks = list(model.get_named_block_list().items())
    gammas, per_block = [], []
    for block_name, block in blocks:
        assert isinstance(block, mb.InvertedResidualChannels)
        bns = block.get_depthwise_bn()
        per_block.append(len(bns))
        gammas.extend(bn.weight for bn in bns)
    if ema is not None:
        ema.attach(runtime.manager_of(model))
    from atomnas_amd import ops
    all_masks, all_index, kept = prune.alive_masks(gammas, threshold, mode=0 if ema is None else (2 if ema_only else 1), with_index=True)
    all_masks = [m.clone() for m in all_masks]   # the arenas are rebuilt while the blocks are compressed
    # Job-list repack: the masks' kept-channel indices and counts are known from that one launch (one device -> host copy for the counts);
    # every gather of the shrink -- weights, BatchNorm vectors, RMSprop state, EMA shadows (models/compress_utils.py:31-37,
    # utils/rmsprop.py:134-165, utils/optim.py:134-153) -- is recorded by the per-tensor protocol and runs as ONE launch at the end
    for m, idx, k in zip(all_masks, all_index, kept.tolist() if kept is not None else []):
        ops.register_mask(m, idx, k)
    ops.gather_defer(True)
    try:
        pos = 0
        for (block_name, block), n in zip(blocks, per_block):
            masks = all_masks[pos:pos + n]
            pos += n
            block.compress_by_mask(masks, ema=ema, optimizer=optimizer, prune_info=prune_info, prefix=block_name, verbose=False)
    finally:
        ops.gather_defer(False)   # flush
        ops.clear_masks()
    if optimizer is not None:
        assert set(id(p) for p in optimizer.param_groups[0]['params']) == set(id(p) for p in model.parameters())
    mc.profiling(model)
    logging.info('Model Shrink to FLOPS: {}'.format(model.n_macs))
    logging.info('Current model: {}'.format(mb.output_network(model)))


def validate(epoch, model_wrapper, ema, criterion, meters, steps):
    """EMA model -> BN calibration (cumulative statistics) -> evaluation (train.py:419-458), on synthetic batches."""
    import common as mc
    FLAGS = cfg.FLAGS
    eval_wrapper = mc.get_ema_model(ema, model_wrapper)
    model = mc.unwrap_model(eval_wrapper)
    if FLAGS.get('bn_calibration', False):
        model.eval()
        model.apply(bn_calibration)
        with torch.no_grad():
            calib = (device_batches(LOADERS[1], FLAGS.bn_calibration_steps) if LOADERS is not None and LOADERS[1] is not None else
                     fake_batches(FLAGS.bn_calibration_per_gpu_batch_size, FLAGS.image_size, FLAGS.model_kwparams['num_classes'],
                                  FLAGS.bn_calibration_steps, 7 + epoch))
            for x, y in calib:
                model(x)
        if FLAGS.use_distributed:
            udist.allreduce_bn(model)
    model.eval()
    with torch.no_grad():
        val = (device_batches(LOADERS[2], steps) if LOADERS is not None and LOADERS[2] is not None else
               fake_batches(FLAGS.per_gpu_batch_size, FLAGS.image_size, FLAGS.model_kwparams['num_classes'], steps, 11))
        for x, y in val:
            mc.forward_loss(model, criterion, x, y, meters)
    return meters.flush(), eval_wrapper


def load_pretrained(path, model_wrapper, ema):
    """`pretrained:` of the reference (train.py:276-296): EMA dict (if any) and model weights of a checkpoint written by
    utils/common.py:123-137 (or a bare state_dict), optionally remapped by position onto this model's keys."""
    FLAGS = cfg.FLAGS
    ckpt = torch.load(path, map_location='cpu', weights_only=False)
    if ema and isinstance(ckpt, dict) and ckpt.get('ema'):
        ema.load_state_dict(ckpt['ema'])
        ema.to(next(model_wrapper.parameters()).device)
    if isinstance(ckpt, dict) and 'model' in ckpt:
        ckpt = ckpt['model']
    if FLAGS.get('pretrained_model_remap_keys', False):
        ckpt = {new: ckpt[old] for new, old in zip(model_wrapper.state_dict().keys(), ckpt.keys())}
    from atomnas_amd.utils.common import unwrap_state_dict
    target = set(model_wrapper.state_dict().keys())
    if not any(k in target for k in ckpt):   # saved through the other wrapper form (`module.` prefix present / absent)
        stripped = unwrap_state_dict(ckpt, verbose=False)
        ckpt = stripped if any(k in target for k in stripped) else {'module.' + k: v for k, v in ckpt.items()}
    model_wrapper.load_state_dict(ckpt)
    logging.info('Loaded model {}.'.format(path))


def load_checkpoint(ckpt, model_wrapper, optimizer, ema):
    """`resume:` of the reference (train.py:299-317) on a checkpoint dict in the reference's format (utils/common.py:123-137:
    'model', 'optimizer' -- torch's index-ordered optimizer state_dict --, 'ema' = {info, shadow, param}, 'last_epoch',
    'best_val').  Checkpoints written here carry the optimizer's parameter ORDER by name as well: after a shrink it differs from
    the model's (re-keyed variables are appended, utils/rmsprop.py:134-165) and torch maps saved state by position."""
    target = set(model_wrapper.state_dict().keys())
    sd = ckpt['model']
    if not any(k in target for k in sd):
        sd = ({k[len('module.'):]: v for k, v in sd.items()} if all(k.startswith('module.') for k in sd)
              else {'module.' + k: v for k, v in sd.items()})
    model_wrapper.load_state_dict(sd)
    names = ckpt.get('optimizer_param_names')
    if names:
        table = dict(model_wrapper.named_parameters())
        if not any(n in table for n in names):   # written under the other wrapper form: same normalisation as the model keys
            names = ([n[len('module.'):] for n in names] if all(n.startswith('module.') for n in names)
                     else ['module.' + n for n in names])
        optimizer.param_groups[0]['params'] = [table[n] for n in names]
    optimizer.load_state_dict(ckpt['optimizer'])
    if ema:
        ema.load_state_dict(ckpt['ema'])
        ema.to(next(model_wrapper.parameters()).device)
    return ckpt['last_epoch'], ckpt['best_val']


def train_val_test():
    import common as mc
    FLAGS = cfg.FLAGS
    model, model_wrapper = mc.get_model()
    ema = mc.setup_ema(model)
    if FLAGS.get('pretrained', None):
        load_pretrained(FLAGS.pretrained, model_wrapper, ema)
    optimizer = optim.get_optimizer(model_wrapper, FLAGS)
    lr_scheduler = optim.get_lr_scheduler(optimizer, FLAGS)
    last_epoch, best_val = -1, 1.0
    FLAGS._global_step = 0
    if FLAGS.resume:
        ckpt = torch.load(os.path.join(FLAGS.resume, 'latest_checkpoint.pt'), map_location='cpu', weights_only=False)
        last_epoch, best_val = load_checkpoint(ckpt, model_wrapper, optimizer, ema)
        lr_scheduler.last_epoch = (last_epoch + 1) * FLAGS._steps_per_epoch
        FLAGS._global_step = (last_epoch + 1) * FLAGS._steps_per_epoch
    assert FLAGS.profiling, '`m.macs` is used for calculating penalty'
    mc.profiling(model)
    FLAGS._bn_to_prune = prune.get_bn_to_prune(model, FLAGS.prune_params, verbose=udist.is_master())
    rho_scheduler = prune.get_rho_scheduler(FLAGS.prune_params, FLAGS._steps_per_epoch)
    step = engine.TrainStep(model, optimizer, ema, FLAGS._bn_to_prune, weight_decay=FLAGS.weight_decay,
                            wd_method=FLAGS.weight_decay_method, label_smoothing=FLAGS.label_smoothing,
                            batch_size=FLAGS.per_gpu_batch_size, image_size=FLAGS.image_size,
                            world_size=udist.get_world_size_fallback(),
                            allreduce_bn=bool(FLAGS.use_distributed and FLAGS.get('allreduce_bn', False)))
    val_criterion = optim.CrossEntropyLabelSmooth(FLAGS.model_kwparams['num_classes'], 0.0, reduction='none')
    val_meters = mc.get_meters('val')
    steps_per_epoch = FLAGS.get('max_steps_per_epoch', None) or FLAGS._steps_per_epoch
    rank = udist.get_rank_fallback()
    for epoch in range(last_epoch + 1, FLAGS.num_epochs):
        model.train()
        t0, seen = time.time(), 0
        batches = (device_batches(LOADERS[0], steps_per_epoch) if LOADERS is not None else
                   fake_batches(FLAGS.per_gpu_batch_size, FLAGS.image_size, FLAGS.model_kwparams['num_classes'], steps_per_epoch,
                                FLAGS.get('random_seed', 0) + rank + 1000 * epoch))
        for x, y in batches:
            if x.shape[0] != FLAGS.per_gpu_batch_size:
                continue   # a short last batch: the captured step has a static batch (drop_last: True avoids drawing it)
            step.set_batch(x, y)
            step.global_step = FLAGS._global_step
            step.step(lr=optimizer.param_groups[0]['lr'], rho=rho_scheduler(FLAGS._global_step))
            lr_scheduler.step()   # (allreduce_bn, when configured, happens inside the step: before the EMA, as in the reference)
            FLAGS._global_step += 1
            seen += x.shape[0]
            if FLAGS._global_step % FLAGS.log_interval == 0 and udist.is_master():
                ce, l2, l1 = step.loss.tolist()   # the only host synchronisation of the interval
                dt = time.time() - t0
                logging.info('Epoch {}/{} step {} loss {:.4f} l2 {:.4f} l1 {:.4f} lr {:.5f} {:.0f} img/s/GPU'.format(
                    epoch, FLAGS.num_epochs, FLAGS._global_step, ce, l2, l1, optimizer.param_groups[0]['lr'], seen / dt))
        results, eval_wrapper = validate(epoch, model_wrapper, ema, val_criterion, val_meters, FLAGS.get('val_steps', 4))
        if udist.is_master():
            logging.info('Epoch {} val: {}'.format(epoch, results))
        if FLAGS.prune_params['method'] is not None:
            thr = FLAGS.model_shrink_threshold
            eval_model = mc.unwrap_model(eval_wrapper)
            masks = prune.cal_mask_network_slimming_by_threshold(get_params_by_name(eval_model, FLAGS._bn_to_prune.weight), thr)
            FLAGS._bn_to_prune.add_info_list('mask', masks)
            flops_pruned, infos = prune.cal_pruned_flops(FLAGS._bn_to_prune)
            if udist.is_master():
                logging.info('Prune threshold: {}, flops pruned: {}, flops remain: {}'.format(thr, flops_pruned, model.n_macs - flops_pruned))
            if flops_pruned >= FLAGS.model_shrink_delta_flops or epoch == FLAGS.num_epochs - 1:
                shrink_model(model_wrapper, ema, optimizer, FLAGS._bn_to_prune, thr, ema_only=(epoch == FLAGS.num_epochs - 1))
        if udist.is_master() and FLAGS.get('log_dir', None):
            os.makedirs(FLAGS.log_dir, exist_ok=True)
            kw = mb.output_network(model)
            pname = {id(p): n for n, p in model_wrapper.named_parameters()}
            state = {'model': {k: v.detach().clone() for k, v in model_wrapper.state_dict().items()}, 'optimizer': optimizer.state_dict(),
                     'optimizer_param_names': [pname[id(p)] for p in optimizer.param_groups[0]['params']],
                     'ema': ema.state_dict() if ema else None, 'last_epoch': epoch, 'best_val': min(best_val, results['top1_error']),
                     # the reference's resume unpacks `train_meters, val_meters = checkpoint['meters']` (train.py:313): a pair
                     'meters': (None, None)}
            torch.save(state, os.path.join(FLAGS.log_dir, 'latest_checkpoint.pt'))
            with open(os.path.join(FLAGS.log_dir, 'latest_checkpoint.yml'), 'w') as f:
                f.write(str(kw))
            if results['top1_error'] < best_val:
                best_val = results['top1_error']
                torch.save(state, os.path.join(FLAGS.log_dir, 'best_model.pt'))
                with open(os.path.join(FLAGS.log_dir, 'best_model.yml'), 'w') as f:
                    f.write(str(kw))


def main():
    import common as mc
    FLAGS = cfg.load_app(sys.argv[1:])
    logging.basicConfig(stream=sys.stdout, level=logging.INFO, format='%(asctime)s %(message)s')
    global LOADERS
    num_train = NUM_IMAGENET_TRAIN
    sets = None
    if FLAGS.get('dataset', 'imagenet1k_fake') != 'imagenet1k_fake':
        # the reference's three factories (train.py:330-339 / utils/dataflow.py:92-267) over a decoded source
        from atomnas_amd.utils import dataflow
        if FLAGS.get('bn_calibration', False):
            FLAGS._loader_batch_size_calib = FLAGS.bn_calibration_per_gpu_batch_size
        sets = dataflow.dataset(*dataflow.data_transforms(FLAGS), FLAGS)
        if sets[0] is not None:
            num_train = len(sets[0])
    mc.setup_distributed(num_train)
    if sets is not None:
        LOADERS = dataflow.data_loader(*sets, FLAGS)
    if udist.is_master():
        logging.info(FLAGS)
    set_random_seed(FLAGS.get('random_seed', 0))
    train_val_test()


if __name__ == '__main__':
    main()
