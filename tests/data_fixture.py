"""A synthetic HO3D-v3-shaped directory tree for the loader tests: frames, masks, per-frame annotations, the three
parameter pickles and OBJ meshes, laid out as HOIG_HOv3/data/hov3_dataset.py:164-257 reads them.  PNG frames (lossless: the tests do
not depend on a JPEG decoder)."""
import os
import pickle
from types import SimpleNamespace

import numpy as np

OBJNAMES = ['003_cracker_box', '004_sugar_box', '006_mustard_bottle', '010_potted_meat_can', '011_banana', '021_bleach_cleanser',
            '025_mug', '035_power_drill', '037_scissors']


def build(root, seed=0, seqs=(('ABF1', 2), ('MC2', 5)), frames=4, frame_hw=(480, 640), mask_hw=(240, 320), n_obj_verts=None):
    """-> opt namespace for HOv3Dataset / CustomDatasetDataLoader.  seqs: (sequence name, object id); videos are '<seq>_0'."""
    from PIL import Image
    g = np.random.Generator(np.random.Philox(key=[seed, 4]))
    pics, params, objs = os.path.join(root, 'images'), os.path.join(root, 'params'), os.path.join(root, 'obj')
    os.makedirs(params, exist_ok=True)
    bbx, vids = {}, {}
    for seq, obj_id in seqs:
        for sub in ('rgb', 'mask', 'meta'):
            os.makedirs(os.path.join(pics, 'train', seq, sub), exist_ok=True)
        vid = seq + '_0'
        bbx[vid] = [float(v) for v in (g.uniform(60, 200), g.uniform(40, 120), g.uniform(180, 330), g.uniform(180, 330))]
        vids[vid] = []
        name = OBJNAMES[obj_id]
        nv = (n_obj_verts or {}).get(obj_id, 50 + 7 * obj_id)
        os.makedirs(os.path.join(objs, name), exist_ok=True)
        with open(os.path.join(objs, name, name + '.obj'), 'w') as f:
            f.write('# synthetic\nmtllib x.mtl\n')
            for v in g.uniform(-0.08, 0.08, (nv, 3)):
                f.write('v %.6f %.6f %.6f\n' % tuple(v))
            f.write('vt 0.5 0.5\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\n')
        for k in range(frames):
            fid = '%04d' % k
            # smooth + noisy content so that interpolation is exercised everywhere
            yy, xx = np.mgrid[0:frame_hw[0], 0:frame_hw[1]]
            base = (np.stack([xx * 0.37 + yy * 0.11, yy * 0.41, (xx + yy) * 0.23], -1) + 40 * k) % 256
            img = np.clip(base + g.integers(-20, 20, base.shape), 0, 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(pics, 'train', seq, 'rgb', fid + '.png'))
            m = (g.uniform(size=mask_hw + (1,)) < 0.3) * np.array([0, 128, 255], np.uint8)        # R channel (cv2's last) = 0 / 255
            blob = np.zeros(mask_hw + (3,), np.uint8)
            blob[mask_hw[0] // 4:mask_hw[0] // 2, mask_hw[1] // 3:mask_hw[1] // 2] = (255, 0, 0)
            Image.fromarray(np.maximum(m.astype(np.uint8), blob)).save(os.path.join(pics, 'train', seq, 'mask', '%05d.png' % k))
            anno = {'objName': name, 'camMat': np.array([[600., 0, 128], [0, 600., 128], [0, 0, 1]]),
                    'handPose': g.standard_normal(48) * 0.3, 'handBeta': g.standard_normal(10), 'handTrans': np.array([0.0, 0.0, -0.5]) + g.uniform(-0.05, 0.05, 3),
                    'objRot': g.uniform(-1.5, 1.5, (3, 1)), 'objTrans': np.array([0.05, 0.0, -0.5]) + g.uniform(-0.02, 0.02, 3)}
            with open(os.path.join(pics, 'train', seq, 'meta', fid + '.pkl'), 'wb') as f:
                pickle.dump(anno, f)
            vids[vid].append(fid + '.png')
    with open(os.path.join(params, 'HOv3-CR_bbx.pkl'), 'wb') as f:
        pickle.dump(bbx, f)
    for n in ('HOv3-CR_train_new.pkl', 'HOv3-CR_test_new.pkl'):
        with open(os.path.join(params, n), 'wb') as f:
            pickle.dump(vids, f)
    return SimpleNamespace(data_dir=root, params_dir='params', images_dir='images', pairs_dir=os.path.join(root, 'pairs.pkl'), obj_dir=objs,
                           dataset_mode='hov3', batch_size=2, serial_batches=True, n_threads_train=0, n_threads_test=0, num_repeats=1)


def write_pairs(opt, pairs):
    with open(opt.pairs_dir, 'wb') as f:
        pickle.dump(pairs, f)


def oracle_batch(opt, names_a, names_b):
    """The reference's collated batch for the named samples, by the CPU oracle (oracle/data_oracle.py) from the files on disk."""
    from PIL import Image
    from oracle import data_oracle as O

    def view(names):
        bbx = pickle.load(open(os.path.join(opt.data_dir, opt.params_dir, 'HOv3-CR_bbx.pkl'), 'rb'))
        out = {k: [] for k in ('image', 'mask', 'cam', 'trans', 'pose', 'shape', 'handtrans', 'vertices_obj', 'objName')}
        for name in names:
            vid, fid = name.split('/')
            seq = vid.split('_')[0]
            base = os.path.join(opt.data_dir, opt.images_dir, 'train', seq)
            bgr = lambda p: np.ascontiguousarray(np.asarray(Image.open(p).convert('RGB'))[:, :, ::-1])
            image, mask, trans = O.sample_tensors(bgr(os.path.join(base, 'rgb', fid)),
                                                  bgr(os.path.join(base, 'mask', '%05d.png' % int(fid.split('.')[0]))), bbx[vid])
            anno = pickle.load(open(os.path.join(base, 'meta', fid.split('.')[0] + '.pkl'), 'rb'))
            v = O.read_obj_vertices(open(os.path.join(opt.obj_dir, anno['objName'], anno['objName'] + '.obj')).read())
            out['image'].append(image); out['mask'].append(mask); out['trans'].append(trans)
            out['cam'].append(anno['camMat'].astype(np.float32)); out['pose'].append(anno['handPose'].astype(np.float32))
            out['shape'].append(anno['handBeta'].astype(np.float32)); out['handtrans'].append(anno['handTrans'].astype(np.float32))
            out['vertices_obj'].append(O.posed_object_vertices(v, anno['objRot'], anno['objTrans']))
            out['objName'].append(OBJNAMES.index(anno['objName']))
        return {k: np.stack(v) for k, v in out.items()}
    return view(names_a), view(names_b)


YCB_NAMES = ['002_master_chef_can', '003_cracker_box', '004_sugar_box', '005_tomato_soup_can', '006_mustard_bottle',
             '007_tuna_fish_can', '008_pudding_box', '009_gelatin_box', '010_potted_meat_can', '011_banana',
             '019_pitcher_base', '021_bleach_cleanser', '024_bowl', '025_mug', '035_power_drill', '036_wood_block',
             '037_scissors', '040_large_marker', '051_large_clamp', '052_extra_large_clamp', '061_foam_brick']


def build_ycb(root, seed=0, vids=(('20200709-subject-01/20200709_141754/836212060125', (3, 11, 16), 1),
                                  ('20200813-subject-02/20200813_145612/932122062010', (7, 2), 0)), frames=3, n_obj_verts=None):
    """A DexYCB-shaped tree for ycb_dataset.py:230-305.  vids: (video id, ycb class ids (1-based), ycb_grasp_ind).  The first video's label
    files carry an all-zero pose BEFORE the grasped object's, so that the reference's index-among-non-zero-poses quirk shows."""
    from PIL import Image
    g = np.random.Generator(np.random.Philox(key=[seed, 9]))
    params = os.path.join(root, 'params')
    os.makedirs(params, exist_ok=True)
    bbx, info, lists = {}, {}, {}
    for vi, (vid, ycb_ids, grasp) in enumerate(vids):
        os.makedirs(os.path.join(root, 'images', vid), exist_ok=True)
        x0, y0 = g.uniform(80, 200), g.uniform(40, 140)
        bbx[vid] = [x0, y0, x0 + g.uniform(180, 300), y0 + g.uniform(180, 300)]
        info[vid] = {'intrinsics': {'fx': 615.0 + vi, 'fy': 614.5, 'ppx': 312.25, 'ppy': 241.5}, 'ycb_grasp_ind': grasp,
                     'ycb_ids': list(ycb_ids), 'mano_betas': [float(v) for v in g.standard_normal(10)]}
        lists[vid] = list(range(frames))
        for cid in ycb_ids:
            name = YCB_NAMES[cid - 1]
            os.makedirs(os.path.join(root, 'models', name), exist_ok=True)
            with open(os.path.join(root, 'models', name, 'textured_pre.obj'), 'w') as f:
                for v in g.uniform(-0.1, 0.1, ((n_obj_verts or {}).get(cid - 1, 40 + 3 * cid), 3)):
                    f.write('v %.6f %.6f %.6f\n' % tuple(v))
                f.write('f 1 2 3\n')
        for k in range(frames):
            yy, xx = np.mgrid[0:480, 0:640]
            img = (np.stack([xx * 0.3 + 20 * k, yy * 0.5, (xx + 2 * yy) * 0.2], -1) % 256).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(root, 'images', vid, 'color_%06d.jpg' % k), quality=92)
            pose_y = np.zeros((len(ycb_ids), 3, 4), np.float32)
            for o in range(len(ycb_ids)):
                if vi == 0 and o == 0:
                    continue                                    # an object without a pose in front of the grasped one
                a = g.uniform(-1, 1, 3)
                th = np.linalg.norm(a)
                K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]) / th
                pose_y[o, :, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
                pose_y[o, :, 3] = g.uniform(-0.3, 0.3, 3) + (0, 0, 0.8)
            pose_m = g.standard_normal((1, 51)).astype(np.float32) * 0.3
            np.savez(os.path.join(root, 'images', vid, 'labels_%06d.npz' % k), pose_y=pose_y, pose_m=pose_m)
    for n, obj in (('DexYCB-bbx.pkl', bbx), ('valid_video_info.pkl', info), ('DexYCB_train.pkl', lists), ('DexYCB_test.pkl', lists)):
        with open(os.path.join(params, n), 'wb') as f:
            pickle.dump(obj, f)
    return SimpleNamespace(data_dir=root, params_dir='params', images_dir='images', pairs_dir=os.path.join(root, 'pairs.pkl'),
                           dataset_mode='ycb', batch_size=2, serial_batches=True, n_threads_train=0, n_threads_test=0, num_repeats=1)


def oracle_batch_ycb(opt, names):
    """The DexYCB copy's collated view for the named samples ('<video id>/<frame>'), by the CPU oracle."""
    from PIL import Image
    from oracle import data_oracle as O
    par = lambda n: pickle.load(open(os.path.join(opt.data_dir, opt.params_dir, n), 'rb'))
    bbx, info = par('DexYCB-bbx.pkl'), par('valid_video_info.pkl')
    out = {k: [] for k in ('image', 'cam', 'trans', 'pose', 'shape', 'vertices_obj', 'objName')}
    for name in names:
        vid, fid = name.rsplit('/', 1)
        bgr = np.ascontiguousarray(np.asarray(Image.open(os.path.join(opt.data_dir, opt.images_dir, vid, 'color_%06d.jpg' % int(fid))).convert('RGB'))[:, :, ::-1])
        image, trans = O.ycb_sample_tensors(bgr, bbx[vid])
        s = info[vid]
        label = np.load(os.path.join(opt.data_dir, 'images', vid, 'labels_%06d.npz' % int(fid)))
        gname = YCB_NAMES[s['ycb_ids'][s['ycb_grasp_ind']] - 1]
        v = O.read_obj_vertices(open(os.path.join(opt.data_dir, 'models', gname, 'textured_pre.obj')).read())
        out['image'].append(image); out['trans'].append(trans)
        out['cam'].append(np.array([s['intrinsics'][k] for k in ('fx', 'fy', 'ppx', 'ppy')], np.float32))
        out['pose'].append(label['pose_m'][0].astype(np.float32)); out['shape'].append(np.array(s['mano_betas'], np.float32))
        out['vertices_obj'].append(O.ycb_object_vertices(v, label['pose_y'], s['ycb_grasp_ind']))
        out['objName'].append(YCB_NAMES.index(gname))
    return {k: np.stack(v) for k, v in out.items()}
