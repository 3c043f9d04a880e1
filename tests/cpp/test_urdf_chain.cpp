// test_urdf_chain.cpp -- host-only: ModelClient::fromURDFString (the part of kdl_parser::treeFromString + the standing-link
// look-up of leg_estimate.cpp:68-73,443-444 the leg kinematics needs) and LegOdoHandler's configuration keys
// (leg_estimate.cpp:29-142, rbis_legodo_update.cpp:10-60).  No GPU, no oracle call.  Exit code 0 + "PASS".
#include <cstdio>
#include <string>

#include "../../pronto_amd/csrc/mav_state_est_batch.hpp"

using namespace MavStateEst;

static const char *URDF = R"(<?xml version="1.0" ?>
<robot name="r" xmlns:xacro="http://ros.org/wiki/xacro">
  <!-- a commented-out joint must not be read: <joint name="ghost" type="revolute"><parent link="base"/><child link="thigh"/></joint> -->
  <link name="base"/>
  <joint name='hip' type='continuous'>
    <origin rpy='0 0 1.5707963' xyz="0.1 -0.2 0.3"/>
    <parent link="base"/> <child link="thigh"/>
    <axis xyz="0 0 -2"/>
  </joint>
  <joint name="knee" type="revolute"><parent link="thigh"/><child link="shin"/><origin xyz="0 0 -0.4"/><axis xyz="0 1 0"/>
    <limit lower="0" upper="2.3" effort="100" velocity="10"/><safety_controller k_velocity="10"/></joint>
  <joint name="slider" type="prismatic"><parent link="shin"/><child link="rod"/><axis xyz="0 0 1"/></joint>
  <joint name="sole_fixed" type="fixed"><parent link="rod"/><child link="l_sole"/><origin xyz="0.01 0 -0.05" rpy="0 0.1 0"/></joint>
  <joint name="r_hip" type="revolute"><parent link="base"/><child link="r_thigh"/></joint>
  <joint name="r_float" type="floating"><parent link="r_thigh"/><child link="r_weird"/></joint>
  <joint name="arm" type="revolute"><parent link="base"/><child link="hand"/><origin xyz="0 0 1"/></joint>
  <gazebo reference="knee"><joint_properties damping="1"/></gazebo>
  <transmission name="t"><joint name="knee"/><actuator name="m"/></transmission>
</robot>)";

int main()
{
  int bad = 0;
  auto expect = [&](bool ok, const char *what) {
    if (!ok) { printf("FAIL: %s\n", what); bad++; }
  };
  std::vector<ModelClient::Joint> ch;
  expect(ModelClient::chainTo(URDF, "l_sole", ch) && ch.size() == 4, "chain base -> l_sole has four joints");
  if (ch.size() == 4) {
    expect(ch[0].name == "hip" && ch[0].type == 1 && ch[0].xyz[1] == -0.2 && ch[0].rpy[2] == 1.5707963 && ch[0].axis[2] == -2.0, "hip: continuous, single quotes, origin, axis");
    expect(ch[1].name == "knee" && ch[1].type == 1 && ch[1].xyz[2] == -0.4 && ch[1].rpy[0] == 0.0 && ch[1].axis[1] == 1.0, "knee: default rpy");
    expect(ch[2].name == "slider" && ch[2].type == 2 && ch[2].xyz[0] == 0.0 && ch[2].axis[2] == 1.0, "slider: prismatic, default origin");
    expect(ch[3].name == "sole_fixed" && ch[3].type == 0 && ch[3].rpy[1] == 0.1 && ch[3].axis[0] == 1.0, "fixed joint, URDF default axis");
  }
  expect(ModelClient::chainTo(URDF, "hand", ch) && ch.size() == 1 && ch[0].name == "arm", "another branch of the tree");
  expect(!ModelClient::chainTo(URDF, "base", ch), "the root link has no chain");
  expect(!ModelClient::chainTo(URDF, "nowhere", ch), "unknown link");
  expect(!ModelClient::chainTo(URDF, "r_weird", ch), "a floating joint on the way is refused");
  {  // XML allows white space around '='; an attribute that is there but does not hold three numbers is an error
    const char *spaced = "<robot><joint name = \"j\" type\t=\n'revolute'><parent link = \"a\"/><child link= \"b\"/>"
                         "<origin xyz = \"1 2 3\" rpy =' 0.1  0.2 0.3 '/><axis xyz\n=\"0 0 1\"/></joint></robot>";
    expect(ModelClient::chainTo(spaced, "b", ch) && ch.size() == 1 && ch[0].xyz[1] == 2.0 && ch[0].rpy[2] == 0.3 && ch[0].axis[2] == 1.0 && ch[0].axis[0] == 0.0,
           "white space around '=' in attributes");
    const char *bad_xyz = "<robot><joint name=\"j\" type=\"revolute\"><parent link=\"a\"/><child link=\"b\"/><origin xyz=\"1 2\"/></joint></robot>";
    expect(!ModelClient::chainTo(bad_xyz, "b", ch), "an <origin xyz> with two numbers is refused");
    const char *bad_axis = "<robot><joint name=\"j\" type=\"revolute\"><parent link=\"a\"/><child link=\"b\"/><axis xyz=\"0 0 one\"/></joint></robot>";
    expect(!ModelClient::chainTo(bad_axis, "b", ch), "an <axis xyz> that is not numeric is refused");
    const char *four = "<robot><joint name=\"j\" type=\"revolute\"><parent link=\"a\"/><child link=\"b\"/><origin rpy=\"0 0 0 0\"/></joint></robot>";
    expect(!ModelClient::chainTo(four, "b", ch), "an <origin rpy> with four numbers is refused");
    ModelClient mb;
    expect(!mb.fromURDFString(bad_xyz, "b", "b"), "fromURDFString fails on an unreadable URDF");
  }
  ModelClient m;
  expect(m.fromURDFString(URDF, "l_sole", "r_thigh") && m.left_chain.size() == 4 && m.right_chain.size() == 1 && m.getURDFString() == URDF, "fromURDFString");
  expect(!m.fromURDFString(URDF, "l_sole", "r_weird"), "fromURDFString with an unsupported chain");

  // the handler reads leg_estimate's and its own keys in its constructor, like the reference
  BotParam param;
  param.applyOverrides("state_estimator.legodo.mode=lin_rate|state_estimator.legodo.r_xyz=0.2|state_estimator.legodo.r_vxyz=0.1|"
                       "state_estimator.legodo.r_vang=0.3|state_estimator.legodo.r_vxyz_uncertain=0.5|state_estimator.legodo.r_vang_uncertain=0.9|"
                       "state_estimator.legodo.zero_initial_velocity=7|state_estimator.legodo.initialization_mode=zero|"
                       "state_estimator.legodo.filter_joint_positions=none|state_estimator.legodo.torque_adjustment=true|"
                       "state_estimator.legodo.adjustment_joints=[\"hip\", \"knee\"]|state_estimator.legodo.adjustment_gain=[7000, 1e4]");
  m.fromURDFString(URDF, "l_sole", "r_thigh");
  LegOdoHandler h(&param, &m);
  expect(h.zero_initial_velocity == 7 && !h.force_torque_init_ && h.use_torque_adjustment_, "handler keys");
  expect(h.adjustment_joints_.size() == 2 && h.adjustment_joints_[1] == "knee" && h.adjustment_gain_.size() == 2 && h.adjustment_gain_[1] == 1e4f, "adjustment arrays");
  msgs::six_axis_force_torque_array_t ft;
  const double fz[2] = { -812.5, 30.25 };
  ft.utime = 1;
  ft.force_z = BatchArray(fz, PB_HOST_BROADCAST);
  h.forceTorqueHandler(&ft, 4);
  expect(h.force_torque_init_ && h.foot_force_.size() == 2 && h.foot_force_[0] == 812.5f && h.foot_force_mem_ == PB_HOST_BROADCAST, "force/torque handler takes |f_z| as float");
  msgs::controller_foot_contact_t cc{ 2, 4, 1 };
  h.controllerInputHandler(&cc);
  expect(h.n_control_contacts_[0] == 4 && h.n_control_contacts_[1] == 1 && h.control_contacts_dirty_, "controller contacts");
  printf(bad ? "FAIL\n" : "PASS\n");
  return bad ? 1 : 0;
}
