//! Scene tree -> the tables of `rt_scene_desc` (include/rtcore.h).  Walks `scene.objects()` depth-first AFTER
//! `compute_bvh()`; child order is kept (it decides ties, group.rs:171-201).  Same walk, same tables as
//! raytracer_amd/host/scene_capi.cpp::flatten_object in the MI355X repository.
use std::collections::HashMap;

use super::rtcore_sys::*;
use crate::math::{color::Color, matrix::Matrix, tuple::Tuple};
use crate::scene::{
    camera::Camera,
    object::{
        bounding_box::Bounded, csg::{CsgObject, CsgOperation}, group::ObjectGroup,
        material::{pattern::Pattern, Material},
        primitive::shape::Shape, Object, ObjectKind, PrimitiveObject,
    },
    Scene,
};

#[derive(Default)]
pub struct FlatScene {
    node_bbox: Vec<f64>,
    node_children: Vec<Vec<u32>>,
    node_child_begin: Vec<u32>,
    child_ref: Vec<u32>,
    prim_shape: Vec<u8>,
    prim_material: Vec<u32>,
    prim_xform: Vec<u32>,
    prim_geom: Vec<u32>,
    prim_csg_left: Vec<u32>,
    xform_inv: Vec<f64>,
    tri: [Vec<f64>; 6], // p1, e1, e2, n1, n2, n3
    quadric_y_min: Vec<f64>,
    quadric_y_max: Vec<f64>,
    quadric_closed: Vec<u8>,
    csg_op: Vec<u8>,
    csg_left: Vec<u32>,
    csg_right: Vec<u32>,
    csg_depth: Vec<u8>,
    material: Vec<f64>,
    pattern_kind: Vec<u8>,
    pattern_c1: Vec<f64>,
    pattern_c2: Vec<f64>,
    pattern_inv: Vec<f64>,
    light_pos: Vec<f64>,
    light_intensity: Vec<f64>,
    material_index: HashMap<Vec<u64>, u32>,
}

fn push3<T: Tuple>(v: &mut Vec<f64>, t: T) { // tuple.rs:21-23
    v.extend_from_slice(&[t.x(), t.y(), t.z()]);
}
fn push_color(v: &mut Vec<f64>, c: Color) { // color.rs:45-53
    v.extend_from_slice(&[c.r(), c.g(), c.b()]);
}
fn matrix16(m: &Matrix) -> [f64; 16] { // row-major through Index<(row, col)>, matrix.rs:262-270
    let mut out = [0.; 16];
    for r in 0..4 {
        for c in 0..4 {
            out[r * 4 + c] = m[(r, c)];
        }
    }
    out
}
const IDENTITY: [f64; 16] = [1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1., 0., 0., 0., 0., 1.];

impl FlatScene {
    pub fn new(scene: &Scene) -> Self {
        let mut f = FlatScene::default();
        let mut chain: Vec<&CsgObject> = Vec::new();
        f.group(scene.objects(), &mut chain); // scene.rs:22; node 0 = the root group
        f.node_child_begin.push(0);
        for kids in std::mem::take(&mut f.node_children) {
            f.child_ref.extend_from_slice(&kids);
            f.node_child_begin.push(f.child_ref.len() as u32);
        }
        for l in scene.light_sources() { // scene.rs:26, point_light.rs:29-35
            push3(&mut f.light_pos, l.position());
            push_color(&mut f.light_intensity, l.intensity());
        }
        f
    }

    fn group<'a>(&mut self, g: &'a ObjectGroup, chain: &mut Vec<&'a CsgObject>) -> u32 {
        let idx = self.node_children.len() as u32;
        self.node_children.push(Vec::new());
        let bb = g.bounding_box(); // group.rs:22-26; min/max are pub fields (bounding_box.rs:25-28)
        push3(&mut self.node_bbox, bb.min);
        push3(&mut self.node_bbox, bb.max);
        let refs: Vec<u32> = g.children().iter().map(|c| self.object(c, chain)).collect(); // group.rs:206, in order
        self.node_children[idx as usize] = refs;
        idx
    }

    fn object<'a>(&mut self, o: &'a Object, chain: &mut Vec<&'a CsgObject>) -> u32 {
        match o.kind() { // object.rs:319
            ObjectKind::Group(g) => rt_make_ref(RT_REF_NODE, self.group(g, chain)),
            ObjectKind::Csg(csg) => {
                assert!(chain.len() < 32, "CSG nesting deeper than 32");
                let idx = self.csg_op.len();
                self.csg_op.push(match csg.operation { // csg.rs:29-34
                    CsgOperation::Union => 0,
                    CsgOperation::Intersection => 1,
                    CsgOperation::Difference => 2,
                });
                self.csg_depth.push(chain.len() as u8);
                self.csg_left.push(0);
                self.csg_right.push(0);
                chain.push(csg.as_ref());
                let l = self.object(&csg.left, chain); // csg.rs:69-74 (pub fields)
                let r = self.object(&csg.right, chain);
                chain.pop();
                self.csg_left[idx] = l;
                self.csg_right[idx] = r;
                rt_make_ref(RT_REF_CSG, idx as u32)
            }
            ObjectKind::Primitive(p) => {
                let idx = self.prim_shape.len() as u32;
                self.primitive(p);
                // csg.rs:93: hit_left = self.left.includes(i.object()), one bit per enclosing CSG object
                let mut mask = 0u32;
                for (d, c) in chain.iter().enumerate() {
                    if c.left.includes(o) { // object.rs:98-104
                        mask |= 1 << d;
                    }
                }
                self.prim_csg_left.push(mask);
                rt_make_ref(RT_REF_PRIM, idx)
            }
        }
    }

    fn primitive(&mut self, p: &PrimitiveObject) {
        let (tag, is_tri) = match p.shape() { // object.rs:403, shape.rs:15-32
            Shape::Sphere => (0u8, false),
            Shape::Plane => (1, false),
            Shape::Cube => (2, false),
            Shape::Cylinder(_) => (3, false),
            Shape::Cone(_) => (4, false),
            Shape::Triangle(_) => (5, true),
            Shape::SmoothTriangle(_) => (6, true),
            Shape::Bbox => (7, false),
        };
        self.prim_shape.push(tag);
        let material = self.material_id(p.material()); // object.rs:441
        self.prim_material.push(material);
        let inv = matrix16(&p.transformation_inverse()); // object.rs:409-411 (None reads as the identity)
        if is_tri || inv.iter().zip(IDENTITY.iter()).all(|(a, b)| a.to_bits() == b.to_bits()) {
            self.prim_xform.push(RT_NONE);
        } else {
            self.prim_xform.push((self.xform_inv.len() / 16) as u32);
            self.xform_inv.extend_from_slice(&inv);
        }
        match p.shape() {
            Shape::Triangle(t) => { // triangle.rs:81-103; the flat normal goes to n1, n2 = n3 = 0
                self.prim_geom.push((self.tri[0].len() / 3) as u32);
                push3(&mut self.tri[0], t.p1());
                push3(&mut self.tri[1], t.e1());
                push3(&mut self.tri[2], t.e2());
                push3(&mut self.tri[3], t.normal());
                self.tri[4].extend_from_slice(&[0., 0., 0.]);
                self.tri[5].extend_from_slice(&[0., 0., 0.]);
            }
            Shape::SmoothTriangle(t) => { // smooth_triangle.rs:93-123
                self.prim_geom.push((self.tri[0].len() / 3) as u32);
                push3(&mut self.tri[0], t.p1());
                push3(&mut self.tri[1], t.e1());
                push3(&mut self.tri[2], t.e2());
                push3(&mut self.tri[3], t.n1());
                push3(&mut self.tri[4], t.n2());
                push3(&mut self.tri[5], t.n3());
            }
            Shape::Cylinder(c) => self.quadric(c.y_min, c.y_max, c.closed), // cylinder.rs:14-18
            Shape::Cone(c) => self.quadric(c.y_min, c.y_max, c.closed),     // cone.rs:13-17
            _ => self.prim_geom.push(RT_NONE),
        }
    }

    fn quadric(&mut self, y_min: f64, y_max: f64, closed: bool) {
        self.prim_geom.push(self.quadric_y_min.len() as u32);
        self.quadric_y_min.push(y_min);
        self.quadric_y_max.push(y_max);
        self.quadric_closed.push(closed as u8);
    }

    /// Materials are shared by value: identical bits -> one table entry (material.rs:10-20, pattern.rs:7-37).
    fn material_id(&mut self, m: &Material) -> u32 {
        let black = Color::new(0., 0., 0.);
        let (kind, c1, c2, inv) = match m.pattern() { // material.rs:43
            Pattern::Const(c) => (0u8, *c, black, IDENTITY),
            Pattern::Stripe { c1, c2, inv_transform } => (1, *c1, *c2, matrix16(inv_transform)),
            Pattern::Gradient { c_start, c_dist, inv_transform } => (2, *c_start, *c_dist, matrix16(inv_transform)),
            Pattern::Ring { c1, c2, inv_transform } => (3, *c1, *c2, matrix16(inv_transform)),
            Pattern::Checkers { c1, c2, inv_transform } => (4, *c1, *c2, matrix16(inv_transform)),
            Pattern::TestPattern { inv_transform } => (5, black, black, matrix16(inv_transform)),
        };
        let scalars = [m.ambient, m.diffuse, m.specular, m.shininess, m.reflectivity, m.transparency, m.refractive_index, 0.];
        let mut key: Vec<u64> = scalars.iter().map(|v| v.to_bits()).collect();
        key.extend([c1.r(), c1.g(), c1.b(), c2.r(), c2.g(), c2.b()].iter().map(|v| v.to_bits()));
        key.extend(inv.iter().map(|v| v.to_bits()));
        key.push(kind as u64);
        if let Some(&id) = self.material_index.get(&key) {
            return id;
        }
        let id = self.pattern_kind.len() as u32;
        self.material.extend_from_slice(&scalars);
        self.pattern_kind.push(kind);
        push_color(&mut self.pattern_c1, c1);
        push_color(&mut self.pattern_c2, c2);
        self.pattern_inv.extend_from_slice(&inv);
        self.material_index.insert(key, id);
        id
    }

    /// The borrowed view handed to `rt_render`; valid while `self` is alive and unchanged.
    pub fn desc(&self) -> rt_scene_desc {
        fn p<T>(v: &[T]) -> *const T {
            if v.is_empty() { std::ptr::null() } else { v.as_ptr() }
        }
        rt_scene_desc {
            abi_version: RT_ABI_VERSION,
            n_nodes: (self.node_child_begin.len() - 1) as u32,
            node_bbox: p(&self.node_bbox),
            node_child_begin: p(&self.node_child_begin),
            n_child_refs: self.child_ref.len() as u32,
            child_ref: p(&self.child_ref),
            n_prims: self.prim_shape.len() as u32,
            prim_shape: p(&self.prim_shape),
            prim_material: p(&self.prim_material),
            prim_xform: p(&self.prim_xform),
            prim_geom: p(&self.prim_geom),
            prim_csg_left: p(&self.prim_csg_left),
            n_xforms: (self.xform_inv.len() / 16) as u32,
            xform_inv: p(&self.xform_inv),
            n_tris: (self.tri[0].len() / 3) as u32,
            tri_p1: p(&self.tri[0]),
            tri_e1: p(&self.tri[1]),
            tri_e2: p(&self.tri[2]),
            tri_n1: p(&self.tri[3]),
            tri_n2: p(&self.tri[4]),
            tri_n3: p(&self.tri[5]),
            n_quadrics: self.quadric_y_min.len() as u32,
            quadric_y_min: p(&self.quadric_y_min),
            quadric_y_max: p(&self.quadric_y_max),
            quadric_closed: p(&self.quadric_closed),
            n_csgs: self.csg_op.len() as u32,
            csg_op: p(&self.csg_op),
            csg_left: p(&self.csg_left),
            csg_right: p(&self.csg_right),
            csg_depth: p(&self.csg_depth),
            n_materials: self.pattern_kind.len() as u32,
            material: p(&self.material),
            pattern_kind: p(&self.pattern_kind),
            pattern_c1: p(&self.pattern_c1),
            pattern_c2: p(&self.pattern_c2),
            pattern_inv: p(&self.pattern_inv),
            n_lights: (self.light_pos.len() / 3) as u32,
            light_pos: p(&self.light_pos),
            light_intensity: p(&self.light_intensity),
            // no device-resident meshes in the plain `image` path (the animate path of hook.rs fills these)
            n_meshes: 0,
            mesh: std::ptr::null(),
            mesh_bbox: std::ptr::null(),
            mesh_built: std::ptr::null(),
            mesh_material_begin: std::ptr::null(),
            mesh_material: std::ptr::null(),
        }
    }
}

/// `Camera` -> `rt_camera`.  half_width / half_height / pixel_size are private there: recomputed exactly as
/// Camera::with_inverse_transformation does (camera.rs:166-175).
pub fn camera_desc(c: &Camera) -> rt_camera {
    let (w, h) = (c.target_width(), c.target_height()); // camera.rs:207-213
    let half_view = (c.field_of_view() / 2.).tan();
    let aspect_ratio = w as f64 / h as f64;
    let (half_width, half_height) = if aspect_ratio >= 1. {
        (half_view, half_view / aspect_ratio)
    } else {
        (half_view * aspect_ratio, half_view)
    };
    rt_camera {
        width: w as u32,
        height: h as u32,
        inv_view: matrix16(&c.inverse_transformation()), // camera.rs:219
        half_width,
        half_height,
        pixel_size: 2. * half_width / w as f64,
    }
}
