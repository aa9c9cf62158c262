"""Train / evaluate the second-stage objectness + refinement network on MI355X.

Command-line drop-in for the reference's ``nerf_rpn/run_rpn_detect.py`` (flags, defaults and choices of run_rpn_detect.py:29-133; side-effect
files ``<save_root>/<process_root>/<process_name>/{epoch_N.pt, model_best_ap25.pt, model_best_ap50.pt, eval.json, objectness/<thr>/*.npz}``;
checkpoint keys ``epoch, backbone_state_dict, RCNN_dict, train_args, optimizer_state_dict, scheduler_state_dict``).  Underneath: HIP kernels
for the backbone, the RoI <-> GT IoU matrices, rotated 3D RoIAlign and the head convolutions (``model/detector.py``).  torch.optim.AdamW +
OneCycleLR are kept so optimizer / scheduler state dicts stay interchangeable with the reference's checkpoints.  ``--use_cuda`` selects the
RoIAlign kernel; without it ROIPool runs the reference's default torch pooling (bit-exact restatement, model/detector.py).
"""
import argparse
import glob
import json
import logging
import math
import os

import numpy as np
import torch
import torch.distributed as dist
from torch.optim import AdamW
from torch.optim.lr_scheduler import OneCycleLR
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from .datasets import RPNClassificationDataset
from .eval import evaluate_box_proposals_ap
from .model.detector import Classification_Model, ProposalTargetLayer, RCNN, ROIPool
from .model.feature_extractor import Bottleneck, ResNet_FPN_256, SwinTransformer_FPN, VGG_FPN
from .model.utils import clip_boxes_to_mesh, nms, remove_small_boxes


def build_parser():
    p = argparse.ArgumentParser(description='Train and eval the NeRF RPN baseline.')
    p.add_argument('--mode', default='train', choices=['train', 'eval'])
    p.add_argument('--debug_mode', action='store_true', help='Turn to debug mode.')
    p.add_argument('--features_path', default='', help='The path to the features.')
    p.add_argument('--boxes_path', default='', help='The path to the boxes.')
    p.add_argument('--rois_path', default='', help='The path to the rois.')
    p.add_argument('--save_root', default='', help='The root to save the model.')
    p.add_argument('--save_path', default='', help='The path to save the model. It will create a folder named the process_name under save_root')
    p.add_argument('--dataset_split', default='', help='The dataset split to use.')
    p.add_argument('--checkpoint', default='', help='The path to the checkpoint to load.')
    p.add_argument('--pretrained', default='', help='The path to the pretrained backbone to load.')
    p.add_argument('--bash_file', default='', help='The bash to run the code.')
    p.add_argument('--fine_tune', action='store_true', help='Fine-tune the backbone.')
    p.add_argument('--backbone_type', type=str, default='resnet', choices=['resnet', 'vgg_AF', 'vgg_EF', 'swin'], help='Backbone type.')
    p.add_argument('--backbone_input_dim', type=int, default=4, help='Input dimension for backbone.')
    p.add_argument('--resolution', type=int, default=160, help='The max resolution of the input features.')
    p.add_argument('--normalize_density', action='store_true', help='Whether to normalize the density.')
    p.add_argument('--output_proposals', action='store_true', help='Whether to output proposals during evaluation.')
    p.add_argument('--filter', choices=['none', 'tp', 'fp'], default='none', help='Filter the proposal output for visualization and debugging.')
    p.add_argument('--filter_threshold', type=float, default=0.5, help='The IoU threshold for the proposal filter, only used if --output_proposals is True and --filter is not "none".')
    p.add_argument('--batch_size', default=2, type=int, help='The num of scenes in a batch.')
    p.add_argument('--num_epochs', default=100, type=int, help='The number of epochs to train.')
    p.add_argument('--lr', default=1e-4, type=float, help='The learning rate.')
    p.add_argument('--reg_loss_weight', default=5.0, type=float, help='The weight for balancing the regression loss.')
    p.add_argument('--weight_decay', default=0.0005, type=float, help='The weight decay coefficient of AdamW.')
    p.add_argument('--clip_grad_norm', default=0.1, type=float, help='The gradient clipping norm.')
    p.add_argument('--rotate_prob', default=0.5, type=float, help='The probability of rotating the scene.')
    p.add_argument('--flip_prob', default=0.5, type=float, help='The probability of flipping the scene.')
    p.add_argument('--rot_scale_prob', default=0.5, type=float, help='The probability of extra scaling and rotation.')
    p.add_argument('--log_interval', default=20, type=int, help='The number of iterations to print the loss.')
    p.add_argument('--eval_interval', default=1, type=int, help='The number of epochs to evaluate.')
    p.add_argument('--keep_checkpoints', default=1, type=int, help='The number of latest checkpoints to keep.')
    p.add_argument('--wandb', action='store_true', help='Whether to use wandb for logging.')
    p.add_argument('--process_root', default='wandb_root', type=str, help='The root of the process in wandb and model save.')
    p.add_argument('--process_name', default='wandb_process', type=str, help='The name of the process.')
    p.add_argument('--init_method', default='tcp://127.0.0.1:23441', type=str, help='init method of wandb.')
    p.add_argument('--gpus', default='', help='The gpus to use for distributed training. If empty, uses the first available gpu. DDP is only enabled if this is greater than one.')
    p.add_argument('--n_classes', default=2, type=int, help='Number of classes for the classification network')
    p.add_argument('--output_size', nargs='+', type=int)
    p.add_argument('--spatial_scale', nargs='+', type=int)
    p.add_argument('--feature_input_dim', default=256, type=int, help='The input dimension of the classification network')
    p.add_argument('--obj_only', action='store_true', help='If true, only train the objectness score.')
    p.add_argument('--enlarge_scale', default=0.2, type=float, help='Control the enlarged ratio of roi')
    p.add_argument('--use_cuda', action='store_true', help='RoI features from the rotated RoIAlign kernel instead of the torch pooling paths')
    p.add_argument('--remap', action='store_true', help='re-map rois to different level')
    p.add_argument('--is_add_layer', action='store_true', help='Add an additional layer to the RCNN')
    p.add_argument('--feature_extracting_type', default='pooling', choices=['pooling', 'interpolation'])
    p.add_argument('--nms_thresh', default=0.1, type=float, help='Parameter for the nms during the evaluation')
    p.add_argument('--filter_score_threhold', default=0.5, type=float, help='During the evaluation, filter out the bounding boxes whose scores are lower than the threshold')
    p.add_argument('--filter_num_threhold', default=300, type=float, help='During the evaluation, filter out the bounding boxes whose scores are lower than the threshold')
    p.add_argument('--cls_batch_size', default=512, type=int, help='batch size of the classification network')
    p.add_argument('--fg_fraction', default=0.5, type=float, help='During the training, the fraction of foreground bounding boxes')
    p.add_argument('--fg_threshold', default=0.35, type=float, help='The threshold of foreground bounding boxes')
    p.add_argument('--bg_threshold', default=0.15, type=float, help='The threshold of background bounding boxes')
    p.add_argument('--top_k', default=None, type=int, help='The top k proposals to compute AP')
    p.add_argument('--rotated_bbox', action='store_true', help='If true, bbox: (N, 7), [x, y, z, w, h, d, theta]. If false, bbox: (N, 6), [xmin, ymin, zmin, xmax, ymax, zmax]')
    p.add_argument('--is_flatten', action='store_true', help='If true, we flatten the features after roi pooling. Otherwise, we take the average of them')
    p.add_argument('--log_to_file', action='store_true', help='Whether to log to a file.')
    p.add_argument('--output_all', action='store_true')
    return p


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def sync_module_from_rank0(module, group=None):
    """Broadcast rank 0's parameters and buffers to every rank (one flat buffer per dtype: a few large transfers instead of one
    collective per tensor)."""
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, ts in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=0, group=group)
        off = 0
        with torch.no_grad():
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()


def allreduce_mean_gradients(params, world_size, group=None):
    """Mean of the gradients over the ranks with ONE all-reduce of a flattened bucket (parameters without a gradient on this rank
    contribute zeros, so every rank reduces the same layout)."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
    dist.all_reduce(flat, group=group)
    flat /= world_size
    off = 0
    for p in params:
        g = flat[off:off + p.numel()].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += p.numel()


class Trainer:
    def __init__(self, args, rank=0, world_size=1, device_id=None, logger=None):
        self.args, self.rank, self.world_size, self.device_id = args, rank, world_size, device_id
        self.logger = logger if logger is not None else logging.getLogger()
        if logger is None:
            self.logger.setLevel(logging.INFO)
        self.logger.info('Constructing model.')
        self.backbone = None
        if args.fine_tune:
            self.build_backbone()
        self.sample_model = ProposalTargetLayer(args.n_classes, batch_size=args.cls_batch_size, fg_fraction=args.fg_fraction,
                                                fg_threshold=args.fg_threshold, bg_threshold=args.bg_threshold, is_rotated_bbox=args.rotated_bbox)
        self.pooling_model = ROIPool(args.output_size, args.spatial_scale, args.enlarge_scale, is_rotated_bbox=args.rotated_bbox,
                                     feature_extracting_type=args.feature_extracting_type, max_res=args.resolution, remap=args.remap,
                                     use_cuda=args.use_cuda)
        self.RCNN_model = RCNN(args.feature_input_dim, Bottleneck, args.n_classes, args.output_size, is_add_layer=args.is_add_layer,
                               is_rotated_bbox=args.rotated_bbox, is_flatten=args.is_flatten)
        self.min_size = 1e-3
        if args.pretrained and args.fine_tune:
            assert os.path.exists(args.pretrained), 'The pretrained model does not exist.'
            self.logger.info(f'Loading pretrained backbone from {args.pretrained}.')
            self.backbone.load_state_dict(torch.load(args.pretrained, map_location='cpu')['backbone_state_dict'])
        if args.checkpoint:
            assert os.path.exists(args.checkpoint), 'The checkpoint does not exist.'
            self.logger.info(f'Loading checkpoint from {args.checkpoint}.')
            ck = torch.load(args.checkpoint, map_location='cpu')
            self.RCNN_model.load_state_dict(ck['RCNN_dict'])
            if self.backbone is not None and ck.get('backbone_state_dict') is not None:
                self.backbone.load_state_dict(ck['backbone_state_dict'])
        self.model = Classification_Model(self.backbone, self.sample_model, self.pooling_model, self.RCNN_model, n_classes=args.n_classes,
                                          is_training=args.mode == 'train', batch_size=args.batch_size, is_rotated_bbox=args.rotated_bbox).cuda()
        if self.world_size > 1:
            # DDP construction semantics (the reference wraps the model, run_rpn_detect.py): every rank starts from rank 0's parameters
            # and buffers -- without this each rank would keep its own random RCNN initialisation and the averaged gradients would be
            # applied to diverged replicas
            sync_module_from_rank0(self.model)
        self.init_datasets()

    def init_datasets(self):
        a = self.args
        if not a.dataset_split:
            raise ValueError('The dataset split must be specified.')
        with np.load(a.dataset_split) as split:
            self.train_scenes, self.test_scenes, self.val_scenes = split['train_scenes'], split['test_scenes'], split['val_scenes']
            if a.output_all:
                self.test_scenes = np.concatenate([self.train_scenes, self.test_scenes, self.val_scenes])
        common = dict(fine_tune=a.fine_tune, normalize_density=a.normalize_density)
        if a.mode == 'eval':
            self.test_set = RPNClassificationDataset(a.features_path, a.boxes_path, a.rois_path, scene_names=self.test_scenes, **common)
            self.logger.info(f'{len(self.test_set)} testing scenes, ')
        else:
            self.val_set = RPNClassificationDataset(a.features_path, a.boxes_path, a.rois_path, scene_names=self.val_scenes, **common)
            self.logger.info(f'{len(self.val_set)} validation scenes.')

    def build_backbone(self):
        a = self.args
        if a.backbone_type == 'resnet':
            self.backbone = ResNet_FPN_256(Bottleneck, [3, 4, 6, 3], input_dim=a.backbone_input_dim, is_max_pool=True)
        elif a.backbone_type in ('vgg_AF', 'vgg_EF'):
            self.backbone = VGG_FPN(a.backbone_type[-2:], a.backbone_input_dim, True, a.resolution)
        else:
            self.backbone = SwinTransformer_FPN(patch_size=[4, 4, 4], embed_dim=96, depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24],
                                                window_size=[4, 4, 4], stochastic_depth_prob=0, expand_dim=True, input_dim=a.backbone_input_dim)

    def save_checkpoint(self, epoch, path):
        torch.save({'epoch': epoch, 'backbone_state_dict': self.backbone.state_dict() if self.args.fine_tune else None,
                    'RCNN_dict': self.RCNN_model.state_dict(), 'train_args': self.args.__dict__,
                    'optimizer_state_dict': self.optimizer.state_dict(), 'scheduler_state_dict': self.scheduler.state_dict()}, path)

    def delete_old_checkpoints(self, path, keep_latest=5):
        files = sorted(glob.glob(f'{path}/epoch_*.pt'), key=os.path.getmtime)
        for f in files[:-keep_latest] if len(files) > keep_latest else []:
            logging.info(f'Deleting old checkpoint {f}.')
            os.remove(f)

    def train_loop(self):
        a = self.args
        self.train_set = RPNClassificationDataset(a.features_path, a.boxes_path, a.rois_path, scene_names=self.train_scenes, fine_tune=a.fine_tune,
                                                  normalize_density=a.normalize_density, rotate_prob=a.rotate_prob, flip_prob=a.flip_prob,
                                                  rotate_scale_prob=a.rot_scale_prob)
        if self.world_size == 1:
            self.train_loader = DataLoader(self.train_set, batch_size=a.batch_size, collate_fn=RPNClassificationDataset.collate_fn, shuffle=True,
                                           num_workers=4, pin_memory=True)
        else:
            self.train_sampler = DistributedSampler(self.train_set)
            self.train_loader = DataLoader(self.train_set, batch_size=a.batch_size // self.world_size, collate_fn=RPNClassificationDataset.collate_fn,
                                           sampler=self.train_sampler, num_workers=2, pin_memory=True)
        self.logger.info(f'{len(self.train_set)} training scenes.')
        self.optimizer = AdamW(self.model.parameters(), lr=a.lr, weight_decay=a.weight_decay)
        self.scheduler = OneCycleLR(self.optimizer, max_lr=a.lr, total_steps=a.num_epochs * len(self.train_loader))
        self.best_metric = None
        start_epoch = 0
        if a.checkpoint:
            ck = torch.load(a.checkpoint, map_location='cpu')
            if 'optimizer_state_dict' in ck:
                self.optimizer.load_state_dict(ck['optimizer_state_dict'])
            if 'scheduler_state_dict' in ck:
                self.scheduler.load_state_dict(ck['scheduler_state_dict'])
            start_epoch = ck['epoch']
        os.makedirs(a.save_path, exist_ok=True)
        for epoch in range(start_epoch, start_epoch + a.num_epochs):
            if self.world_size > 1:
                self.train_sampler.set_epoch(epoch)
            self.train_epoch(epoch)
            if self.rank != 0:
                continue
            if epoch % a.eval_interval == 0 or epoch == a.num_epochs - 1:
                APs, _, _ = self.eval(self.val_set)
                if self.best_metric is None:
                    self.best_metric = APs
                else:
                    for name in list(APs.keys()):
                        if APs[name] > self.best_metric[name]:
                            self.best_metric[name] = APs[name]
                            self.save_checkpoint(epoch, os.path.join(a.save_path, f'model_best_ap{name}.pt'))
                self.save_checkpoint(epoch, os.path.join(a.save_path, f'epoch_{epoch}.pt'))
                self.delete_old_checkpoints(a.save_path, keep_latest=a.keep_checkpoints)

    def _to_device(self, level_features, boxes, rois):
        return ([[t.cuda() for t in item] for item in level_features], [b.cuda().float() for b in boxes], [r.cuda().float() for r in rois])

    def train_epoch(self, epoch):
        a = self.args
        for i, batch in enumerate(self.train_loader):
            self.model.train()
            self.optimizer.zero_grad()
            level_features, boxes, rois, scene_name = batch
            gt_labels = [b.new_ones(b.size(0)) for b in boxes]      # binary objectness: every ground-truth box has label 1
            level_features, boxes, rois = self._to_device(level_features, boxes, rois)
            _, _, losses = self.model(rois, boxes, gt_labels, level_features)
            losses['loss_rpn_box_reg'] = losses['loss_rpn_box_reg'] * a.reg_loss_weight
            loss = losses['loss_objectness'] if a.obj_only else losses['loss_objectness'] + losses['loss_rpn_box_reg']
            loss.backward()
            if self.world_size > 1:       # one process per GPU over RCCL: mean of the gradients (the reference wraps the model in DDP)
                allreduce_mean_gradients(self.model.parameters(), self.world_size)
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), a.clip_grad_norm)
            self.optimizer.step()
            self.scheduler.step()
            if i % a.log_interval == 0 and self.rank == 0:
                self.logger.info(f'Epoch {epoch} [{i}/{len(self.train_loader)}]  Loss: {loss.item():.4f}  '
                                 f'Obj loss: {losses["loss_objectness"].item():.4f}  Reg loss: {losses["loss_rpn_box_reg"].item():.4f}')

    def output_proposals(self, scenes, proposals, scores, gt_boxes, threshold=0.7):
        out = os.path.join(self.args.save_path, 'objectness', f'{math.floor(threshold * 1000)}')
        os.makedirs(out, exist_ok=True)
        for scene, proposal, score in zip(scenes, proposals, scores):
            keep = torch.nonzero(score >= threshold).view(-1)
            np.savez(os.path.join(out, f'{scene}.npz'), proposal=proposal[keep].numpy(), score=score[keep].numpy())

    @torch.no_grad()
    def filter_proposals(self, proposals, objectness, gt_labels, mesh_sizes, score_threhold=0.8):
        size = 7 if self.args.rotated_bbox else 6
        fb, fs, fl = [], [], []
        for boxes, scores, labels, mesh_shape in zip(proposals, objectness, gt_labels, mesh_sizes):
            boxes = clip_boxes_to_mesh(boxes, mesh_shape)
            keep = remove_small_boxes(boxes, self.min_size)
            boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
            keep = torch.where(scores[..., 1] >= score_threhold)[0] if scores.dim() > 1 else torch.where(scores >= score_threhold)[0]
            boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
            keep = nms(boxes[..., :size], scores[..., 1], self.args.nms_thresh)
            keep = keep[scores[..., 1][keep].sort(descending=True)[1]]
            fb.append(boxes[keep][..., :size])
            fs.append(scores[keep])
            fl.append(labels[keep])
        return fb, fs, fl

    @torch.no_grad()
    def eval(self, dataset):
        a = self.args
        self.model.eval()
        loader = DataLoader(dataset, batch_size=1, shuffle=False, num_workers=1, collate_fn=RPNClassificationDataset.collate_fn)
        self.logger.info('Evaluating...')
        props_all, scores_all, gt_all, scenes_all = [], [], [], []
        chunk = max(1, a.cls_batch_size // self.world_size)
        for level_features, boxes, rois, scene_name in loader:
            gt_labels = [b.new_ones(b.size(0)) for b in boxes]
            level_features, boxes, rois = self._to_device(level_features, boxes, rois)
            out_boxes, out_labels, out_scores = [], [], []
            for b in range(len(rois)):
                pb, pl, ps = [], [], []
                for s in range(0, rois[b].size(0), chunk):
                    (cp, cl), cs, _ = self.model([rois[b][s:s + chunk]], [boxes[b]], [gt_labels[b]], [level_features[b]], is_sample=False,
                                                 is_reg=not a.obj_only)
                    pb.append(cp[0]); pl.append(cl[0]); ps.append(cs[0])
                out_boxes.append(torch.cat(pb)); out_labels.append(torch.cat(pl)); out_scores.append(torch.cat(ps))
            mesh_sizes = [[v * a.spatial_scale[0] for v in level_features[i][0].shape[1:]] for i in range(len(level_features))]
            if a.fine_tune:
                mesh_sizes = [list(level_features[i][0].shape[1:]) for i in range(len(level_features))]
            fb, fs, fl = self.filter_proposals(out_boxes, out_scores, out_labels, mesh_sizes, score_threhold=a.filter_score_threhold)
            props_all += [t.cpu() for t in fb]
            scores_all += [t[..., 1].cpu() for t in fs]
            gt_all += [b.cpu() for b in boxes]
            scenes_all += list(scene_name)
        json_dict, APs = {}, {}
        for thr in (0.25, 0.5):
            AP = evaluate_box_proposals_ap(props_all, scores_all, gt_all, thr, top_k=a.top_k)
            self.logger.info(f'AP{int(thr * 100)}: {float(AP["ap"])}')
            json_dict[f'AP{int(thr * 100)}'] = {k: (v.tolist() if isinstance(v, torch.Tensor) else v) for k, v in AP.items()}
            APs[f'{int(thr * 100)}'] = float(AP['ap'])
        if a.mode == 'eval':
            os.makedirs(a.save_path, exist_ok=True)
            path = os.path.join(a.save_path, 'eval.json')
            with open(path, 'a' if os.path.exists(path) else 'w') as f:
                json.dump(json_dict, f, indent=2)
            if a.output_proposals:
                self.output_proposals(scenes_all, props_all, scores_all, gt_all, threshold=a.filter_score_threhold)
        return APs, [], []


def main(argv=None):
    args = parse_args(argv)
    args.save_path = os.path.join(args.save_root, args.process_root, args.process_name)
    os.makedirs(args.save_path, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s %(levelname)s %(message)s')
    logger = logging.getLogger('nerf_rpn_detect')
    if args.log_to_file:
        logger.addHandler(logging.FileHandler(os.path.join(args.save_path, 'log.txt')))
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    if world > 1:
        from .affinity import pin_rank
        pin_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))      # NRPN_PIN=0: off
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
    trainer = Trainer(args, rank, world, local, logger)
    if args.mode == 'train':
        trainer.train_loop()
    else:
        trainer.eval(trainer.test_set)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
