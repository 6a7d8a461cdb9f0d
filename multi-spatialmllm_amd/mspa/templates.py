"""Default question / answer / task-description tables for the task heads.

These are this project's own wordings (a few per list).  The *placeholders* and table keys are the
contract (they are what the reference's records are filled from); the reference's larger tables can
be plugged in unchanged with ``TemplateSet.from_module`` -- any module exposing the same names the
reference's scripts use (``TASK_DESCRIPTION``, ``QUESTION_TEMPLATES`` / ``ANSWER_TEMPLATES`` dicts
keyed by question type, or a ``TEMPLATES`` dict with "questions"/"answers").  With identical tables and
identical ``random`` seeds the heads emit byte-identical records (the heads draw from ``random`` in the
reference's order).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Sequence

_TWO_IMG = "Image-1: <image>\nImage-2: <image>\n"
_COORD_NOTE = ("Coordinates [ x , y ] are normalised to 0-1 and multiplied by 1000, origin [ 0 , 0 ] at the "
               "top-left corner, x along the width and y along the height.")


@dataclasses.dataclass
class TemplateSet:
    task_description: List[str]
    questions: Dict[str, List[str]]      # question type -> templates
    answers: Dict[str, List[str]]

    @staticmethod
    def from_module(mod, question_types: Sequence[str] = ()) -> "TemplateSet":
        # object_perception's script names its list ASK_DESCRIPTION (OPE:25) although it reads TASK_DESCRIPTION (OPE:190)
        task = list(getattr(mod, "TASK_DESCRIPTION", None) or getattr(mod, "ASK_DESCRIPTION"))
        if isinstance(getattr(mod, "QUESTION_TEMPLATES", None), (list, tuple)):
            return TemplateSet(task, {"default": list(mod.QUESTION_TEMPLATES)}, {"default": list(mod.ANSWER_TEMPLATES)})
        if hasattr(mod, "QUESTION_TEMPLATES"):
            return TemplateSet(task, {k: list(v) for k, v in mod.QUESTION_TEMPLATES.items()},
                               {k: list(v) for k, v in mod.ANSWER_TEMPLATES.items()})
        t = getattr(mod, "TEMPLATES")
        keys = list(question_types) or ["default"]
        return TemplateSet(task, {k: list(t["questions"]) for k in keys}, {k: list(t["answers"]) for k in keys})


CAMERA_MOVEMENT_TYPES = ("x_movement", "y_movement", "z_movement", "yaw_movement", "pitch_movement",
                         "total_distance", "yaw_angle", "pitch_angle", "displacement_vector")

CAMERA_MOVEMENT = TemplateSet(
    task_description=[
        _TWO_IMG + "Two photos of a static scene were taken from different camera poses. Describe the camera motion "
                   "from the first pose to the second, expressed in the first camera's frame.",
        _TWO_IMG + "The scene does not change between the two images; only the camera moves. Work out how it moved, "
                   "relative to the first image.",
    ],
    questions={
        "x_movement": ["Did the camera move left or right?", "Which horizontal direction did the camera shift in?"],
        "y_movement": ["Did the camera move up or down?", "Which vertical direction did the camera shift in?"],
        "z_movement": ["Did the camera move forward or backward?", "Along the viewing axis, which way did the camera go?"],
        "yaw_movement": ["Did the camera turn left or right?", "Which way did the camera pan?"],
        "pitch_movement": ["Did the camera tilt up or down?", "Which way did the camera pitch?"],
        "total_distance": ["How far apart (in mm) are the two camera positions?",
                           "What distance in millimetres did the camera travel?"],
        "yaw_angle": ["By how many degrees did the camera pan?", "What is the size of the yaw rotation in degrees?"],
        "pitch_angle": ["By how many degrees did the camera tilt?", "What is the size of the pitch rotation in degrees?"],
        "displacement_vector": ["With X right, Y down and Z forward in the first image, what is the camera's "
                                "displacement vector in mm?",
                                "Axes of the first view: X right, Y down, Z forward. Give the translation of the camera "
                                "as [ x , y , z ] in mm."],
    },
    answers={
        "x_movement": ["It moved `{x_movement}`.", "The shift is to the `{x_movement}`."],
        "y_movement": ["It moved `{y_movement}`.", "The camera went `{y_movement}`."],
        "z_movement": ["It moved `{z_movement}`.", "The camera went `{z_movement}`."],
        "yaw_movement": ["It turned `{yaw_movement}`.", "The pan is to the `{yaw_movement}`."],
        "pitch_movement": ["It tilted `{pitch_movement}`.", "The tilt is `{pitch_movement}`."],
        "total_distance": ["About `{total_distance}` mm.", "The two positions are `{total_distance}` mm apart."],
        "yaw_angle": ["Roughly `{yaw_angle}` degrees.", "The pan measures `{yaw_angle}` degrees."],
        "pitch_angle": ["Roughly `{pitch_angle}` degrees.", "The tilt measures `{pitch_angle}` degrees."],
        "displacement_vector": ["`[ {x_value} , {y_value} , {z_value} ]` mm.",
                                "The translation is `[ {x_value} , {y_value} , {z_value} ]` mm."],
    })

VISUAL_CORRESPONDENCE = TemplateSet(
    task_description=[_TWO_IMG + "Find where a point of the first image appears in the second image. " + _COORD_NOTE,
                      _TWO_IMG + "Match points between the two views. " + _COORD_NOTE],
    questions={"default": ["The point [ {x1} , {y1} ] is marked in Image-1. Where is it in Image-2?",
                           "Locate in the second image the point seen at [ {x1} , {y1} ] in the first."]},
    answers={"default": ["It is at [ {x2} , {y2} ].", "The matching position is [ {x2} , {y2} ]."]})

DEPTH_ESTIMATION = TemplateSet(
    task_description=["<image>\nAnswer a depth question about one point of the image. " + _COORD_NOTE,
                      "<image>\nA pixel is given by its coordinates; report how far it is from the camera. " + _COORD_NOTE],
    questions={"default": ["What is the depth (in mm) at [ {x1} , {y1} ]?",
                           "How far from the camera, in millimetres, is the point [ {x1} , {y1} ]?"]},
    answers={"default": ["`{depth}` mm.", "The point [ {x1} , {y1} ] is `{depth}` mm away."]})

DEPTH_COMPARISON = TemplateSet(
    task_description=["<image>\nTwo points of the image are given by their coordinates; compare their distance to the "
                      "camera. " + _COORD_NOTE,
                      "<image>\nDecide which of two image points lies nearer to or farther from the camera. " + _COORD_NOTE],
    questions={"closer": ["Which of [ {x1} , {y1} ] and [ {x2} , {y2} ] is closer to the camera?",
                          "Of the points [ {x1} , {y1} ] and [ {x2} , {y2} ], which one is nearer?"],
               "farther": ["Which of [ {x1} , {y1} ] and [ {x2} , {y2} ] is farther from the camera?",
                           "Of the points [ {x1} , {y1} ] and [ {x2} , {y2} ], which one is more distant?"]},
    answers={"closer": ["`[ {correct_x} , {correct_y} ]` is closer.", "The nearer point is `[ {correct_x} , {correct_y} ]`."],
             "farther": ["`[ {correct_x} , {correct_y} ]` is farther.",
                         "The more distant point is `[ {correct_x} , {correct_y} ]`."]})

VISUAL_CORRESPONDENCE_DOT = TemplateSet(
    task_description=[_TWO_IMG + "A point is marked with a dot in Image-1; Image-2 shows four lettered candidates. Pick the one "
                                 "that is the same scene point.",
                      _TWO_IMG + "Match the marked point of the first image with one of the lettered dots in the second."],
    questions={"default": ["Which lettered dot in Image-2 corresponds to the dot in Image-1?",
                           "Choose the letter of the point in the second image that matches the marked point."]},
    answers={"default": ["`{correct_label}`", "The matching point is `{correct_label}`."]})

OBJECT_MOVEMENT_DOT = TemplateSet(
    task_description=[_TWO_IMG + "A point is marked with a dot in Image-1; objects and the camera may have moved before Image-2. "
                                 "Describe the motion of the marked point relative to the first image.",
                      _TWO_IMG + "Track the dotted point of the first frame into the second frame."],
    questions={"tapvid3d_total_distance": ["How far (in mm) did the marked point move?",
                                           "Give the travelled distance in mm of the dotted point."],
               "tapvid3d_displacement_vector": ["With X right, Y down, Z forward in the first view, what is the displacement (mm) "
                                                "of the marked point?",
                                                "Report the [ x , y , z ] displacement in mm of the dotted point."]},
    answers={"tapvid3d_total_distance": ["It moved `{total_distance}` mm.", "The distance is `{total_distance}` mm."],
             "tapvid3d_displacement_vector": ["`[ {x_value} , {y_value} , {z_value} ]` mm.",
                                              "Its displacement is `[ {x_value} , {y_value} , {z_value} ]` mm."]})

DEPTH_ESTIMATION_DOT = TemplateSet(
    task_description=["<image>\nOne point of the image is marked with a coloured dot; report its distance from the camera.",
                      "<image>\nEstimate the depth at the marked dot."],
    questions={"default": ["What is the depth (in mm) at the marked point?", "How far from the camera, in millimetres, is the dot?"]},
    answers={"default": ["`{depth}` mm.", "The marked point is `{depth}` mm away."]})

DEPTH_COMPARISON_DOT = TemplateSet(
    task_description=["<image>\nTwo points are marked with lettered dots; compare their distance to the camera.",
                      "<image>\nDecide which lettered dot lies nearer to or farther from the camera."],
    questions={"closer": ["Which marked point is closer to the camera?", "Which lettered dot is nearer to the viewer?"],
               "farther": ["Which marked point is farther from the camera?", "Which lettered dot is more distant?"]},
    answers={"closer": ["Point `{correct_label}` is closer.", "`{correct_label}` is the nearer point."],
             "farther": ["Point `{correct_label}` is farther.", "`{correct_label}` is the more distant point."]})

OBJECT_MOVEMENT_TYPES = ("tapvid3d_total_distance", "tapvid3d_displacement_vector")

OBJECT_MOVEMENT = TemplateSet(
    task_description=[_TWO_IMG + "Objects and the camera may or may not have moved between the two frames. Describe the "
                                 "motion relative to the first image.",
                      _TWO_IMG + "Compare the two frames; either the object, the camera, both or neither moved."],
    questions={
        "tapvid3d_total_distance": ["How far (in mm) did the point at [ {x1} , {y1} ] of Image-1 move? " + _COORD_NOTE,
                                    "Give the travelled distance in mm of the point seen at [ {x1} , {y1} ]. " + _COORD_NOTE],
        "tapvid3d_displacement_vector": ["With X right, Y down, Z forward in the first view, what is the displacement (mm) of "
                                         "the point at [ {x1} , {y1} ]? " + _COORD_NOTE,
                                         "Report the [ x , y , z ] displacement in mm of the point at [ {x1} , {y1} ]. "
                                         + _COORD_NOTE],
    },
    answers={
        "tapvid3d_total_distance": ["It moved `{total_distance}` mm.", "The distance is `{total_distance}` mm."],
        "tapvid3d_displacement_vector": ["`[ {x_value} , {y_value} , {z_value} ]` mm.",
                                         "Its displacement is `[ {x_value} , {y_value} , {z_value} ]` mm."],
    })

OBJECT_PERCEPTION = TemplateSet(
    task_description=["The scene is static. Use every image together to answer a question about an object's size.",
                      "All images show the same unchanged scene; combine them to measure the object."],
    questions={"default": ["What is the {dimension} in millimetres of the {object_category} that these images show?",
                           "Measure the {dimension} (mm) of the {object_category} seen across these images."]},
    answers={"default": ["Its {dimension} is about `{value_mm}` mm.",
                         "The {object_category} has a {dimension} of roughly `{value_mm}` mm."]})
